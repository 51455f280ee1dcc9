"""Not a test: times k_demod alone (dabphy_time_demod: B x F frame slots, oscillator on, f Hz) for quick A/B runs on the GPU box.
usage: python tools/time_demod.py [lib.so ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

torch.cuda.init()
from __graft_entry__ import load_package  # noqa: E402

load_package()
import parity_cases as P  # noqa: E402
from welle_io_amd import capi, synth  # noqa: E402

x = synth.make_stream(5, snr_db=20, seed=1)
frames = P.cut_frames(x, 4)
libs = sys.argv[1:] or [os.path.join(ROOT, "welle.io_amd", "libdabphy_hip.so")]
chunks = [int(c) for c in os.environ.get("CHUNKS", "25").split(",")]
for lib in libs:
    for ch in chunks:
        d = capi.DabPhy(lib_path=lib, demod_chunk=ch)
        # (the first short run only wakes the clocks up: a launch of an idle device takes 2.4 ms, the 300th in a row 1.8)
        iters = int(os.environ.get("ITERS", "300"))
        d.time_demod(frames, 256, 20, mix=1, f_hz=137, iters=100)
        for f in (137, 137, 0):
            print(os.path.basename(lib), "chunk", ch, "f_hz", f, "ms", d.time_demod(frames, 256, 20, mix=1, f_hz=f, iters=iters), flush=True)
        d.close()
