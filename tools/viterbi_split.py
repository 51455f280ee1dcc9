"""Not a test: the structural experiment of round 6 on the throughput Viterbi kernel (profiles/r06_viterbi_split.txt).  The traceback of
k_viterbi_fused as a pass of its own BESIDE the forward pass (k_traceback_fused: 24-register waves, one per SIMD as a sixth wave next to the
five forward waves of 96 registers; the forward waves publish a group's decisions -- agent-scope release -- and go on to the next group;
dabphy_test_traceback_split) against the kernel as it was (every wave walks back the group whose trellis it ran):
  * the decode launch ALONE on the benchmark batch (dabphy_time_fused_msc), both ways;
  * the whole pipelined step (process + superframe filter, HIP-event stage times), both ways, interleaved A/B/A/B;
  * bytes: FIBs, CRC flags and the MSC bytes of three sub-channels of every ensemble of one batch, split against unsplit.
usage: python tools/viterbi_split.py [B] [F]"""
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
lib = os.environ.get("DABPHY_LIB", os.path.join(PKG_DIR, "libdabphy_hip.so"))
rec = workload.rec_frames_for(F)
iq, cfo, base, txs = workload.make_batch(B, rec_frames=rec)
out = {}
for split in (0, 1):
    dev = workload.open_receiver(capi, lib, iq, F, txs[0].subchs)
    dev.traceback_split(bool(split))
    for _ in range(3):
        dev.process(F); dev.superframes_stats()
    torch.cuda.synchronize()
    fib, ok = dev.fibs(); msc = [dev.msc(i)[0].copy() for i in (0, 7, 17)]
    out[split] = (fib.copy(), ok.copy(), msc, dev.superframes_stats().copy())
    print("split %d: decode launch alone %.3f ms (5 launches in a row, nothing else on the device)" % (split, dev.time_fused_msc(5)), flush=True)
    dev.close()
same = np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and all(np.array_equal(a, b) for a, b in zip(out[0][2], out[1][2])) and np.array_equal(out[0][3], out[1][3])
print("FIBs, CRC flags, MSC bytes of sub-channels 0 / 7 / 17 of all %d ensembles and the superframe totals of batch 3: %s" % (B, "equal" if same else "DIFFERENT"), flush=True)
devs = {}
for split in (0, 1):
    devs[split] = workload.open_receiver(capi, lib, iq, F, txs[0].subchs)
    devs[split].traceback_split(bool(split))
    for _ in range(4):
        devs[split].process(F); devs[split].superframes_stats()
for rnd in range(3):
    for split in (0, 1):
        dev = devs[split]
        torch.cuda.synchronize(); t0 = time.perf_counter(); acc = {}
        for _ in range(10):
            dev.process(F); dev.superframes_stats(); dev.fibs_host()
            for k, v in dev.stage_times().items():
                acc[k] = acc.get(k, 0.0) + v
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print("round %d split %d: step %.3f ms = %.0f x real-time; demod %.3f  decode %.3f  filter %.3f  sync chain %.3f" % (rnd, split, dt * 1e3, B * F * 0.096 / dt, acc["demod"] / 10, acc["msc_viterbi"] / 10, acc["rs"] / 10, acc["sync"] / 10), flush=True)
for d in devs.values():
    d.close()
