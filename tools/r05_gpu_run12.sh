#!/bin/bash
# Not a test: round 5, last device session (7 GPU-minutes were left): the profiles of the FINAL sources (tools/make_profiles.sh) and the device
# tests of what changed since the full -m gpu suite of tools/r05_gpu_run8.sh (208 passed): exact batch mode's second pass.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5p; rm -rf $O; mkdir -p $O
bash tools/make_profiles.sh > $O/make_profiles.log 2>&1
tail -n 3 $O/make_profiles.log
timeout 150 python -m pytest tests/test_gpu_stream.py -m gpu -x -q -k "lock_lost or exact_batch or replayed or dropout or relock" > $O/gputest_subset.log 2>&1; echo "pytest rc $?" >> $O/gputest_subset.log
tail -n 3 $O/gputest_subset.log
python - <<'PY'
import json
try:
    j = json.loads([l for l in open("gpurun_out/prof/bench.json").read().splitlines() if l.startswith("{")][-1])
    print("value", j["value"], "ms", j["ms_per_step"], "frac", j["roofline"]["frac"], "stages", j["stages_ms"])
except Exception as e:
    print("no bench line", e); print(open("gpurun_out/prof/bench.err").read()[-2000:])
PY
