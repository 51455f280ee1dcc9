"""Not a test: the pipelined step of the benchmark batch under the round-6 scheduling options, A/B on one box, interleaved rounds
(profiles/r06_step_variants.txt):
  base       the next batch's synchroniser behind the decoder (sync_early = 1), superframe filter at the end of the step (auto = 1)
  early      ... in front of the decoder (dabphy_config.sync_early = 0, the default since round 6)
  deferred   the filter pass of batch k beside batch k + 1's FFT stage (dabphy_set_auto_superframes(2))
  both       early + deferred
  both+split ... and the lane-per-code-word kernel's traceback as a pass of its own (dabphy_test_traceback_split)
usage: python tools/step_variants.py [B] [F] [variant ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package, PKG_DIR  # noqa: E402

load_package()
import torch  # noqa: E402
from welle_io_amd import capi, workload  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
F = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ALL = {"base": dict(sync_early=1, deferred=False, split=False), "early": dict(sync_early=0, deferred=False, split=False),
       "deferred": dict(sync_early=1, deferred=True, split=False), "both": dict(sync_early=0, deferred=True, split=False),
       "both+split": dict(sync_early=0, deferred=True, split=True), "front-wait": dict(sync_early=3, deferred=False, split=False)}
names = sys.argv[3:] or ["base", "early", "deferred", "both"]
lib = os.environ.get("DABPHY_LIB", os.path.join(PKG_DIR, "libdabphy_hip.so"))
iq, cfo, base, txs = workload.make_batch(B, rec_frames=workload.rec_frames_for(F))
devs = {}
for n in names:
    v = ALL[n]
    d = workload.open_receiver(capi, lib, iq, F, txs[0].subchs, sync_early=v["sync_early"], deferred_filter=v["deferred"])
    if v["split"]:
        d.traceback_split(True)
    tot = np.zeros(4, np.int64)
    for _ in range(5):
        d.process(F); tot += d.superframes_stats().sum(0)
    devs[n] = d
    print("%-10s warm: superframe totals of 5 batches %s" % (n, tot.tolist()), flush=True)
for rnd in range(3):
    for n in names:
        d = devs[n]
        torch.cuda.synchronize(); t0 = time.perf_counter(); acc = {}; tot = np.zeros(4, np.int64)
        for _ in range(10):
            d.process(F); tot += d.superframes_stats().sum(0); d.fibs_host()
            for k, v in d.stage_times().items():
                acc[k] = acc.get(k, 0.0) + v
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print("round %d %-10s step %.3f ms = %6.0f x real-time; demod %.3f  decode %.3f  filter %.3f  sync chain %.3f; superframes %d uncorrectable %d" % (
            rnd, n, dt * 1e3, B * F * 0.096 / dt, acc["demod"] / 10, acc["msc_viterbi"] / 10, acc["rs"] / 10, acc["sync"] / 10, tot[0], tot[2]), flush=True)
for d in devs.values():
    d.close()
