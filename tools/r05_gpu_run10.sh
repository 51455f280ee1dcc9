#!/bin/bash
# Not a test: round 5, LAST device session, on the final sources (block-parallel traceback; the history ring put back in exact batch mode):
# what tools/r05_gpu_run8.sh did -- the whole -m gpu suite, tools/make_profiles.sh, the step's time line.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5n; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q --durations=10 > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
tail -n 6 $O/gputest.log
bash tools/make_profiles.sh > $O/make_profiles.log 2>&1
tail -n 3 $O/make_profiles.log
rm -rf gpurun_out/kt; mkdir -p gpurun_out/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -o kt -- python bench.py --no-cpu-baseline --no-alt-schedule --no-extras > $O/kt.log 2>&1
python tools/step_timeline.py $(find gpurun_out/kt -name "kt_kernel_trace.csv" | head -1) > $O/step_timeline.txt 2>&1
cat $O/step_timeline.txt
python - <<'PY'
import json
try:
    j = json.loads([l for l in open("gpurun_out/prof/bench.json").read().splitlines() if l.startswith("{")][-1])
    print("value", j["value"], "ms", j["ms_per_step"], "frac", j["roofline"]["frac"], "stages", j["stages_ms"])
    ex = j.get("extras", {})
    for k in ("mixed_layouts", "hetero"):
        e = ex.get(k, {}); print(k, e.get("value"), e.get("ms_per_step"), e.get("parity"), e.get("error"), e.get("parity_error"))
    print("parity", j.get("parity_check", {}).get("against"), j["cpu_baseline"]["value"])
except Exception as e:
    print("no bench line", e); print(open("gpurun_out/prof/bench.err").read()[-2000:])
PY
