#!/bin/bash
# Not a test: round 4, random parity sweeps on the device with the round's decoders: exact batch mode at 2-8 dB and through random channels,
# 2 ... 8 frames per call, all schedules -- once with the default kernel choice (these batches: state-parallel) and once lane per code
# word (the 144- / 324-row builds of the fused kernel); then the default bench line on the committed profiles.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4e; rm -rf $O; mkdir -p $O
SWEEP_DECODE_SHAPE=0 timeout 900 python tools/sweep_batch.py 70 401 exact > $O/sweep_exact_sp.txt 2> $O/sweep.err; echo "rc $?" >> $O/sweep_exact_sp.txt
SWEEP_DECODE_SHAPE=1 timeout 900 python tools/sweep_batch.py 70 402 exact > $O/sweep_exact_lane.txt 2>> $O/sweep.err; echo "rc $?" >> $O/sweep_exact_lane.txt
SWEEP_DECODE_SHAPE=0 timeout 900 python tools/sweep_batch.py 50 403 channels > $O/sweep_channels_sp.txt 2>> $O/sweep.err; echo "rc $?" >> $O/sweep_channels_sp.txt
SWEEP_DECODE_SHAPE=1 timeout 900 python tools/sweep_batch.py 50 404 channels > $O/sweep_channels_lane.txt 2>> $O/sweep.err; echo "rc $?" >> $O/sweep_channels_lane.txt
tail -n 2 $O/sweep_*.txt; tail -n 5 $O/sweep.err
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r4e/bench.json") if l.startswith("{")][-1])
print("ms_per_step %.3f value %.0f frac %.3f traffic %s vit frac %s stale %s" % (j["ms_per_step"], j["value"], j["roofline"]["frac"], j["roofline"]["traffic"], j["roofline_viterbi"].get("frac"), j["profile_build"]["stale_profile"]))
PY
