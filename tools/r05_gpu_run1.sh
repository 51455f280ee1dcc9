#!/bin/bash
# Not a test: round 5, first device session: the new per-ensemble selection tests, then the whole -m gpu suite, then the default bench line
# (with extras.mixed_layouts).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5a; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_bench_config.py -m gpu -x -q -k "mid_stream or deep_batches or independent_ensembles" > $O/new_tests.log 2>&1; echo "pytest rc $?" >> $O/new_tests.log
tail -n 15 $O/new_tests.log
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
tail -n 25 $O/gputest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
try:
    j = json.loads(open("gpurun_out/r5a/bench.json").read().strip().splitlines()[-1])
    print("value", j["value"], "ms", j["ms_per_step"], "frac", j["roofline"]["frac"], "stages", j["stages_ms"])
    print("mixed", json.dumps(j.get("extras", {}).get("mixed_layouts"))[:1500])
    print("hetero", json.dumps(j.get("extras", {}).get("hetero"))[:600])
except Exception as e:
    print("no bench line", e); print(open("gpurun_out/r5a/bench.err").read()[-2000:])
PY
