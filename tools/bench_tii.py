"""Cost of the TII side path on the bench workload: dabphy_process with and without dabphy_set_tii (HBM-resident looping IQ).
Usage (GPU box): python tools/bench_tii.py [B] [F]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: F401,E402
from conftest import GPU_LIB  # noqa: E402
from welle_io_amd import capi, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
F = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nets = [[(3, 17, 0, 1.0), (11, 40, 37, 0.6)], [(23, 69, 0, 1.0)], None, [(0, 0, 0, 0.8), (1, 0, 12, 0.7)]]
base = [synth.make_stream(F, snr_db=25, seed=70 + i, tii=nets[i]) for i in range(4)]
x = np.stack([base[b % 4] for b in range(B)])
d = capi.DabPhy(lib_path=GPU_LIB, n_ensembles=B, max_frames=F, pipeline_sync=True, want_constellation=False)
d.stream_upload(x, loop=True)
for on in (False, True, False, True):
    d.set_tii(on)
    for _ in range(2):
        d.process(F)
    t = time.time()
    for _ in range(5):
        d.process(F)
    dt = (time.time() - t) / 5
    ev, n = d.tii()
    print("tii=%d  %.2f ms/step  %.0f x real-time  measurements in last batch: %d" % (on, dt * 1e3, B * F * 0.096 / dt, int(n.sum())))
d.close()
