import os, sys, time
ROOT = "/root/repo" if os.path.isdir("/root/repo/tools") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package, PKG_DIR
load_package()
import torch
from welle_io_amd import capi, workload
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
    print("%-34s step %.3f ms; demod %.3f decode %.3f filter %.3f sync %.3f" % (tag, dt * 1e3, acc["demod"] / 10, acc["msc_viterbi"] / 10, acc["rs"] / 10, acc["sync"] / 10), flush=True)
mode = sys.argv[1]
if mode == "tiny":
    t = capi.DabPhy(lib_path=lib, n_ensembles=1, max_frames=1)     # five streams, a few MB
    b = open_(); timeit("behind a TINY handle", b); b.close(); t.close()
elif mode == "streams":
    ss = [torch.cuda.Stream() for _ in range(5)]
    b = open_(); timeit("behind 5 torch streams", b); b.close()
elif mode == "two":
    a = open_(); b = open_(); timeit("B", b); timeit("A", a)
    import ctypes
