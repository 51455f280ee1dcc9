#!/bin/bash
# Not a test: round 4, A/B of the superframe filter's access-unit CRCs (four bytes per step) and of the synchroniser's placement once the filter is shorter
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4j; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_stream.py -m gpu -x -q -k "superframe" > $O/gputest.log 2>&1; tail -n 2 $O/gputest.log
for early in 0 1 0 1; do
  DABPHY_CHAIN_EARLY=$early DABPHY_LIB=$PWD/gpurun_in/lib_exp.so timeout 300 python bench.py --no-cpu-baseline --no-alt-schedule --no-extras --steps 20 > $O/bench_early${early}_$RANDOM.json 2>> $O/early.err
done
rm -rf gpurun_out/kt; mkdir -p gpurun_out/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -o kt -- python bench.py --no-cpu-baseline --no-alt-schedule --no-extras > $O/kt.log 2>&1
python tools/step_timeline.py $(find gpurun_out/kt -name "kt_kernel_trace.csv" | head -1) > $O/step_timeline.txt 2>&1
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r4j/bench*.json")):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(os.path.basename(f), "ms_per_step %.3f value %.0f demod %.3f vit %.3f rs %.3f" % (j["ms_per_step"], j["value"], j["stages_ms"]["demod"], j["stages_ms"]["msc_viterbi"], j["stages_ms"]["rs"]))
PY
cat $O/step_timeline.txt; tail -n 3 $O/early.err
