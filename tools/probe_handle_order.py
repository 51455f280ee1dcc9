"""Not a test: why does the SECOND handle a process opens decode the benchmark batch ~3 % faster than the first (round 6: tools/step_variants.py
showed the step following the handle's creation order, not the option under test)?  Opens handles one after the other -- some closed again, some
behind a large dummy allocation -- and times each (10 steps, stage times).  usage: python tools/probe_handle_order.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package, PKG_DIR  # noqa: E402

load_package()
import torch  # noqa: E402
from welle_io_amd import capi, workload  # noqa: E402

B, F = 256, 32
lib = os.path.join(PKG_DIR, "libdabphy_hip.so")
iq, cfo, base, txs = workload.make_batch(B, rec_frames=workload.rec_frames_for(F))


def open_():
    d = workload.open_receiver(capi, lib, iq, F, txs[0].subchs)
    for _ in range(5):
        d.process(F); d.superframes_stats()
    return d


def timeit(tag, d):
    torch.cuda.synchronize(); t0 = time.perf_counter(); acc = {}
    for _ in range(10):
        d.process(F); d.superframes_stats(); d.fibs_host()
        for k, v in d.stage_times().items():
            acc[k] = acc.get(k, 0.0) + v
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("%-28s step %.3f ms; demod %.3f decode %.3f filter %.3f sync %.3f" % (tag, dt * 1e3, acc["demod"] / 10, acc["msc_viterbi"] / 10, acc["rs"] / 10, acc["sync"] / 10), flush=True)


a = open_(); timeit("A (1st)", a); timeit("A again", a)
b = open_(); timeit("B (2nd, A open)", b); timeit("A (B open)", a)
a.close(); timeit("B (A closed)", b)
c = open_(); timeit("C (3rd, in A's place?)", c); timeit("B (C open)", b)
b.close(); c.close()
d = open_(); timeit("D (alone again)", d)
d.close()
pad = torch.empty(int(12e9), dtype=torch.uint8, device="cuda")
e = open_(); timeit("E (behind a 12 GB dummy)", e)
del pad; torch.cuda.empty_cache()
timeit("E (dummy freed)", e)
e.close()
