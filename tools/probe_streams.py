"""Not a test: the benchmark step and the overlapped bulk drain under different stream layouts of the handle (experiments build,
DABPHY_STREAM_LAYOUT: bit 0 placeholder streams, bit 1 FIC work on the auxiliary stream, bit 3 the drain on a stream of its own), each layout in a
process of its own.  usage: DABPHY_LIB=<experiments build> DABPHY_STREAM_LAYOUT=n python tools/probe_streams.py"""
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

B, F = 256, 32
lib = os.environ.get("DABPHY_LIB", os.path.join(PKG_DIR, "libdabphy_hip.so"))
iq, cfo, base, txs = workload.make_batch(B, rec_frames=workload.rec_frames_for(F))
d = workload.open_receiver(capi, lib, iq, F, txs[0].subchs)
for _ in range(5):
    d.process(F); d.superframes_stats()
tag = "layout %s" % os.environ.get("DABPHY_STREAM_LAYOUT", "default")
for rnd in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); acc = {}
    for _ in range(10):
        d.process(F); d.superframes_stats(); d.fibs_host()
        for k, v in d.stage_times().items():
            acc[k] = acc.get(k, 0.0) + v
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("%s: step %.3f ms; demod %.3f decode %.3f filter %.3f sync %.3f fic %.3f" % (tag, dt * 1e3, acc["demod"] / 10, acc["msc_viterbi"] / 10, acc["rs"] / 10, acc["sync"] / 10, acc["fic"] / 10), flush=True)
if os.environ.get("PROBE_NO_DRAIN"):
    d.close(); sys.exit(0)
nb, nd = d.msc_batch_size()
pinned = d.host_alloc((nb,), np.uint8)
d.msc_batch(pinned)
for rnd in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); acc = {}
    for _ in range(10):
        d.msc_drain_begin(pinned)
        d.process(F); d.superframes_stats(); d.fibs_host()
        d.msc_drain_wait()
        for k, v in d.stage_times().items():
            acc[k] = acc.get(k, 0.0) + v
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("%s: step WITH the drain %.3f ms; demod %.3f decode %.3f filter %.3f sync %.3f" % (tag, dt * 1e3, acc["demod"] / 10, acc["msc_viterbi"] / 10, acc["rs"] / 10, acc["sync"] / 10), flush=True)
d.msc_drain_wait(); d.host_free(pinned); d.close()
