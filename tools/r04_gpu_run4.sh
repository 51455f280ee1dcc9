#!/bin/bash
# Not a test: round 4, profile session on the round's final build: the default bench line, tools/make_profiles.sh (kernel statistics +
# FETCH_SIZE / WRITE_SIZE / SQ counter passes), a kernel trace for the step's time line, the new / changed device tests once more.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4d; rm -rf $O; mkdir -p $O
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
bash tools/make_profiles.sh > $O/make_profiles.log 2>&1
rm -rf gpurun_out/kt; mkdir -p gpurun_out/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -o kt -- python bench.py --no-cpu-baseline --no-alt-schedule --no-extras > $O/kt.log 2>&1
python tools/step_timeline.py $(find gpurun_out/kt -name "kt_kernel_trace.csv" | head -1) > $O/step_timeline.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_parity.py tests/test_gpu_bench_entry.py -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
tail -n 4 $O/gputest.log
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r4d/bench.json") if l.startswith("{")][-1])
print("ms_per_step %.3f value %.0f" % (j["ms_per_step"], j["value"]), j["stages_ms"], "frac %.3f copy %s" % (j["roofline"]["frac"], j["roofline"].get("measured_copy_GBps")))
PY
cat $O/step_timeline.txt; tail -n 6 $O/make_profiles.log
