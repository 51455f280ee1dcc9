#!/bin/bash
# Not a test: round 4, final device session: the whole -m gpu suite, the default bench line (-> profiles/r04_bench_default.json), the
# rocprofv3 passes of tools/make_profiles.sh, a kernel trace for the step's time line, the chain-placement A/B, the decode-shape sweep.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
tail -4 $O/gputest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
for early in 0 1 0 1; do
  DABPHY_CHAIN_EARLY=$early DABPHY_LIB=$PWD/gpurun_in/lib_exp.so timeout 300 python bench.py --no-cpu-baseline --no-alt-schedule --no-extras --steps 20 > $O/bench_early${early}_$RANDOM.json 2>> $O/early.err
done
timeout 600 python tools/sweep_decode_shape.py > $O/sweep_decode_shape.txt 2> $O/sweep.err
timeout 120 tools/ubench/copy_f4 > $O/copy_f4.txt 2>&1
bash tools/make_profiles.sh > $O/make_profiles.log 2>&1
mkdir -p gpurun_out/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -o kt -- python bench.py --no-cpu-baseline --no-alt-schedule --no-extras > $O/kt.log 2>&1
python tools/step_timeline.py $(find gpurun_out/kt -name "kt_kernel_trace.csv" | head -1) > $O/step_timeline.txt 2>&1
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r4c/bench*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), "ms_per_step %.3f value %.0f demod %.3f vit %.3f fic %.3f rs %.3f frac %.3f" % (j["ms_per_step"], j["value"], j["stages_ms"]["demod"], j["stages_ms"]["msc_viterbi"], j["stages_ms"]["fic"], j["stages_ms"]["rs"], j["roofline"]["frac"]),
              "hetero", (j.get("extras") or {}).get("hetero", {}).get("value"), "facade", (j.get("facade") or {}).get("ms_per_frame"))
    except Exception as e:
        print(f, "unreadable", e)
PY
cat $O/sweep_decode_shape.txt $O/step_timeline.txt; tail -n 12 $O/make_profiles.log; tail -n 3 $O/early.err
