#!/bin/bash
# Not a test: round 5, second device session: the whole -m gpu suite on the round's sources so far, the default bench line with every extra,
# the A/B of leaving a few per cent of the decoder's wave slots to the next batch's synchroniser (experiments build), a kernel trace for
# the step's time line.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5b; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q --durations=12 > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
tail -n 22 $O/gputest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
for pct in 100 97 94 100 97 94; do
  DABPHY_VITM_SLOTS_PCT=$pct DABPHY_LIB=$PWD/gpurun_in/lib_exp.so timeout 300 python bench.py --no-cpu-baseline --no-alt-schedule --no-extras --steps 20 > $O/bench_pct${pct}_$RANDOM.json 2>> $O/pct.err
done
rm -rf gpurun_out/kt; mkdir -p gpurun_out/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -o kt -- python bench.py --no-cpu-baseline --no-alt-schedule --no-extras > $O/kt.log 2>&1
python tools/step_timeline.py $(find gpurun_out/kt -name "kt_kernel_trace.csv" | head -1) > $O/step_timeline.txt 2>&1
cat $O/step_timeline.txt
python - <<'PY'
import json, glob
try:
    j = json.loads(open("gpurun_out/r5b/bench.json").read().strip().splitlines()[-1])
    print("value", j["value"], "ms", j["ms_per_step"], "frac", j["roofline"]["frac"], "stages", j["stages_ms"])
    ex = j.get("extras", {})
    for k in ("mixed_layouts", "hetero"):
        e = ex.get(k, {})
        print(k, e.get("value"), e.get("ms_per_step"), e.get("stages_ms"), e.get("parity"), e.get("error"), e.get("parity_error"))
    print("facade", json.dumps(j.get("facade"))[:1200])
    print("short", json.dumps(ex.get("short_batches"))[:800])
    print("cpu", j["cpu_baseline"]["value"], j["cpu_baseline"]["kind"], j.get("parity_check", {}).get("against"))
except Exception as e:
    print("no bench line", e); print(open("gpurun_out/r5b/bench.err").read()[-2000:])
for f in sorted(glob.glob("gpurun_out/r5b/bench_pct*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["ms_per_step"], j["stages_ms"]["msc_viterbi"], j["stages_ms"]["demod"])
    except Exception as e:
        print(f, "failed", e)
PY
