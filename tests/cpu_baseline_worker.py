"""Not a test: one CPU receiver over a looping recording, timed; bench.py starts one of these per host core for its `cpu_baseline`
leg and compares what the first ones decoded with the GPU's output for the same rows (the run's parity check).
argv: recording.npy n_loops mode(port|reference|reference_o3|reference_prof) out.npz|- [subchannels.json]
  port            the oracle = single-threaded C restatement of the reference's PHY
  reference       the real reference backend (oracle/_ref): RadioReceiver with its own threads, FIBProcessor, DecoderAdapter (incl. AAC)
  reference_o3    the same sources built -O3 -march=x86-64-v3 (oracle/Makefile ref-variants: SURVEY 8d's courtesy build)
  reference_prof  ... built with -DWITH_PROFILING: the reference's own PROFILE() marks; that build writes profiling_points.csv into the
                  working directory when the process ends (various/profiling.cpp:119-188)"""
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
from welle_io_amd import synth, workload  # noqa: E402

rec = np.load(sys.argv[1]); n_loops = int(sys.argv[2]); mode = sys.argv[3]; out = sys.argv[4]
if mode in ("reference_o3", "reference_prof"):
    os.environ["WELLE_REF_LIB"] = os.path.join(ROOT, "oracle", "_ref", "libwelle_%s.so" % mode.replace("reference", "ref"))
    mode = "reference"
if len(sys.argv) > 5:
    subchs = workload.subchannels_from_json(json.load(open(sys.argv[5])))    # another multiplex (bench.py's hetero leg)
else:
    subchs = synth.EnsembleTx(eid=0x1000, seed=0).subchs      # the canonical 18 x 64 kbit/s EEP-3A layout
x = np.tile(rec, n_loops)                                     # the looping ring as the device sees it, noise and carrier offset included
if mode == "reference":
    R.ref()
else:
    R.orc(); R.orc_nco_table()                                 # load + build tables outside the timed region
print("READY", flush=True)
sys.stdin.readline()                                           # all receivers start together
t0 = time.time()
if mode == "reference":
    a = R.receiver_run(x, subchs=subchs)
    dt = time.time() - t0
    fib, msc, frames = a["fib"], a["msc"], len(a["fib"]) // 12
else:
    o = R.orc_receiver_run(x, subchs=subchs)
    dt = time.time() - t0
    fib, msc, frames = o["fib"], o["msc"], int(o["n_frames"])
if out != "-":
    keep = 64                                                  # frames of output kept for the parity comparison
    np.savez(out, fib=fib[:12 * keep], **{"msc%d" % i: np.frombuffer(bytes(msc[i])[:4 * keep * subchs[i].frame_bytes], np.uint8) for i in range(len(subchs))})
print(json.dumps({"frames": frames, "seconds": dt, "fib_ok": int(fib[:, 0].sum()), "fibs": int(len(fib))}), flush=True)
