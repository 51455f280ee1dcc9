"""Not a test: runs k_demod back to back for a few seconds while sampling rocm-smi (shader clock, power), to see whether the kernel's
sustained rate is set by the power limit.  usage (GPU box): python tools/clock_probe.py [lib.so]"""
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

torch.cuda.init()
from __graft_entry__ import load_package  # noqa: E402

load_package()
import parity_cases as P  # noqa: E402
from welle_io_amd import capi, synth  # noqa: E402

lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "welle.io_amd", "libdabphy_hip.so")
x = synth.make_stream(5, snr_db=20, seed=1)
frames = P.cut_frames(x, 4)
d = capi.DabPhy(lib_path=lib, demod_chunk=25)
stop = False
samples = []


def poll():
    while not stop:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        keep = [ln.strip() for ln in r.stdout.splitlines() if ("sclk" in ln or "Power" in ln or "mclk" in ln) and "GPU[0]" in ln]
        samples.append((time.time(), keep))
        time.sleep(0.2)


th = threading.Thread(target=poll); th.start()
t0 = time.time()
for it in (1, 2, 5, 20, 100, 300):
    ms = d.time_demod(frames, 256, 20, mix=1, f_hz=137, iters=it)
    print("iters %4d: %.3f ms per launch   (t = %.1f s)" % (it, ms, time.time() - t0), flush=True)
stop = True; th.join()
for t, k in samples:
    print("%.1f s  %s" % (t - t0, " | ".join(k)))
d.close()
