"""Not a test: one single-threaded CPU receiver (the oracle = CPU restatement of the reference algorithm) over a recording,
timed; bench.py starts one of these per host core for its `cpu_baseline` leg.  argv: recording.npy n_loops seed"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402
load_package()
import refapi as R  # noqa: E402
from welle_io_amd import synth  # noqa: E402

base = np.load(sys.argv[1]); n_loops = int(sys.argv[2]); seed = int(sys.argv[3])
use_reference = len(sys.argv) > 4 and sys.argv[4] == "reference"      # the real reference backend (oracle/_ref) instead of the restatement
subchs = synth.EnsembleTx(eid=0x1000, seed=0).subchs          # the canonical 18 x 64 kbit/s EEP-3A layout
x = np.tile(base, n_loops)
rng = np.random.RandomState(seed)
xv = x.view(np.float32)
for i in range(0, len(xv), 1 << 22):
    xv[i:i + (1 << 22)] += (0.02 * rng.randn(len(xv[i:i + (1 << 22)]))).astype(np.float32)
if use_reference:
    R.ref()
else:
    R.orc(); R.orc_nco_table()                                 # load + build tables outside the timed region
print("READY", flush=True)
sys.stdin.readline()                                           # all receivers start together
t0 = time.time()
if use_reference:
    a = R.receiver_run(x, subchs=subchs)                        # RadioReceiver + its own threads, FIBProcessor, DecoderAdapter (incl. AAC)
    dt = time.time() - t0
    print(json.dumps({"frames": len(a["fib"]) // 12, "seconds": dt, "fib_ok": int(a["fib"][:, 0].sum()), "fibs": int(len(a["fib"]))}), flush=True)
else:
    o = R.orc_receiver_run(x, subchs=subchs)
    print(json.dumps({"frames": int(o["n_frames"]), "seconds": time.time() - t0, "fib_ok": int(o["fib"][:, 0].sum()), "fibs": int(len(o["fib"]))}), flush=True)
