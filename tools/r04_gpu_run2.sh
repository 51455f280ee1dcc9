#!/bin/bash
# Not a test: round 4, second device session: the state-parallel kernel on the device (stream / parity / host-mirror suites, which decode
# small batches), the two kernels against each other over batch sizes, the facade's latency, the copy sweep, the default bench line.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4b; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
tail -5 $O/gputest.log
timeout 120 tools/ubench/copy_f4 > $O/copy_f4.txt 2>&1
timeout 600 python tools/sweep_decode_shape.py > $O/sweep_decode_shape.txt 2> $O/sweep.err
timeout 200 python tools/bench_facade.py --json > $O/facade.json 2> $O/facade.err
timeout 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r4b/bench*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), "ms_per_step %.3f value %.0f demod %.3f vit %.3f fic %.3f frac %.3f" % (j["ms_per_step"], j["value"], j["stages_ms"]["demod"], j["stages_ms"]["msc_viterbi"], j["stages_ms"]["fic"], j["roofline"]["frac"]),
              "hetero", (j.get("extras") or {}).get("hetero", {}).get("value"), (j.get("extras") or {}).get("hetero", {}).get("msc_viterbi_ms"), (j.get("extras") or {}).get("hetero", {}).get("parity"),
              "copy", j["roofline"].get("measured_copy_GBps"), j["roofline"].get("measured_copy_note", "")[-120:], "facade", (j.get("facade") or {}).get("ms_per_frame"))
    except Exception as e:
        print(f, "unreadable", e)
PY
cat $O/sweep_decode_shape.txt $O/facade.json; grep -E "read only|write only|contiguous" $O/copy_f4.txt | tail -16; tail -n 3 $O/sweep.err; tail -n 3 $O/bench.err
