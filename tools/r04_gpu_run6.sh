#!/bin/bash
# Not a test: round 4, closing device session on the final build: the whole -m gpu suite, tools/make_profiles.sh, a kernel trace for the
# step's time line, the decode-shape sweep, the default bench line.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4i; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
tail -n 4 $O/gputest.log
bash tools/make_profiles.sh > $O/make_profiles.log 2>&1
rm -rf gpurun_out/kt; mkdir -p gpurun_out/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -o kt -- python bench.py --no-cpu-baseline --no-alt-schedule --no-extras > $O/kt.log 2>&1
python tools/step_timeline.py $(find gpurun_out/kt -name "kt_kernel_trace.csv" | head -1) > $O/step_timeline.txt 2>&1
true
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r4i/bench.json") if l.startswith("{")][-1])
print("ms_per_step %.3f value %.0f" % (j["ms_per_step"], j["value"]), j["stages_ms"], "frac %.3f copy %s" % (j["roofline"]["frac"], j["roofline"].get("measured_copy_GBps")), "hetero", j["extras"]["hetero"].get("value"), j["extras"]["hetero"].get("parity"), "facade", j["facade"].get("ms_per_frame"))
PY
cat $O/step_timeline.txt
