"""How often, and for which ensembles, k_sync_finish needs the ordered float sums in steady lock (GPU box): python tools/probe_exact.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: F401,E402
from conftest import GPU_LIB  # noqa: E402
from welle_io_amd import capi, synth  # noqa: E402

B, F = 64, 20
rng = np.random.RandomState(1)
N = 20 * 196608
base = synth.make_stream(20, snr_db=None, seed=10)[:N]
n = np.arange(N)
xs = []
for b in range(B):
    cfo = np.round(rng.uniform(-60, 60) * N / 2048000.0) * 2048000.0 / N          # phase-continuous when the recording loops
    noise = (rng.randn(N) + 1j * rng.randn(N)).astype(np.complex64) * 0.02
    xs.append((base * np.exp(2j * np.pi * cfo * n / 2048000.0)).astype(np.complex64) + noise)
d = capi.DabPhy(lib_path=GPU_LIB, n_ensembles=B, max_frames=F, want_constellation=False)
d.stream_upload(np.stack(xs), loop=True)
prev = np.zeros(B, np.int64)
for step in range(8):
    d.process(F)
    lost, ex = d.sync_stats()
    delta = ex - prev; prev = ex.copy()
    print("batch", step, "frames settled by ordered sums:", int(delta.sum()), "of", B * F, " ensembles involved:", int((delta > 0).sum()), " per-ensemble counts:", sorted(delta[delta > 0].tolist(), reverse=True)[:12])
d.close()
