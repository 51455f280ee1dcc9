"""Not a test: what k_demod's time is made of (profiles/r06_demod_split.txt).  The kernel alone (dabphy_time_demod: 256 x 32 frame slots, warm
clocks, 200 launches in a row) in the product build -- oscillator off (mix = 0), oscillator on a frame whose frequency is 0 (the table
shortcut), oscillator on (f = 137 Hz: today's hot path) -- and in three experiment builds of the same sources (csrc/Makefile
EXTRA="-DDABPHY_EXPERIMENTS -DDEMOD_EXP_...", wrong results by construction): the double-precision oscillator tree in single precision
(DEMOD_EXP_OSC_F32), the demapper stubbed (DEMOD_EXP_NODEMAP), both.
usage: python tools/demod_split.py            (builds the experiment libraries under /tmp with hipcc, then times all of them)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

torch.cuda.init()
from __graft_entry__ import load_package  # noqa: E402

load_package()
import parity_cases as P  # noqa: E402
from welle_io_amd import capi, synth  # noqa: E402

CSRC = os.path.join(ROOT, "welle.io_amd", "csrc")
BUILDS = [("product", None), ("osc_f32", "-DDEMOD_EXP_OSC_F32"), ("nodemap", "-DDEMOD_EXP_NODEMAP"), ("osc_f32+nodemap", "-DDEMOD_EXP_OSC_F32 -DDEMOD_EXP_NODEMAP")]
libs = {}
for name, flags in BUILDS:
    if flags is None:
        libs[name] = os.path.join(ROOT, "welle.io_amd", "libdabphy_hip.so"); continue
    out = "/tmp/libdabphy_%s.so" % name.replace("+", "_")
    r = subprocess.run(["make", "-j8", "OUT=" + out, "OBJDIR=/tmp/build_" + name.replace("+", "_"), "EXTRA=-DDABPHY_EXPERIMENTS " + flags], cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        print("build", name, "failed:", r.stdout[-800:]); continue
    libs[name] = out
x = synth.make_stream(5, snr_db=20, seed=1)
frames = P.cut_frames(x, 4)
B, F, iters = 256, int(os.environ.get("FRAMES", "32")), int(os.environ.get("ITERS", "200"))
alg = B * F * (76 * 2048 * 8 + 75 * 3072)
print("k_demod alone, %d x %d frame slots, %d launches in a row after a warm-up of 100 (ms per launch; GB/s of algorithmic bytes; fraction of 8 TB/s)" % (B, F, iters))
rows = {}
for name, lib in libs.items():
    d = capi.DabPhy(lib_path=lib, demod_chunk=25)
    d.time_demod(frames, B, F, mix=1, f_hz=137, iters=100)
    for what, mix, f in (("mix=0", 0, 0), ("mix=1 f=0 (table shortcut)", 1, 0), ("mix=1 f=137", 1, 137), ("mix=1 f=137 again", 1, 137)):
        ms = d.time_demod(frames, B, F, mix=mix, f_hz=f, iters=iters)
        rows[(name, what)] = ms
        print("%-18s %-28s %.4f ms  %.0f GB/s  %.3f" % (name, what, ms, alg / ms / 1e6, alg / ms / 1e6 / 8000.0), flush=True)
    d.close()
try:
    p = rows[("product", "mix=1 f=137 again")]
    print("oscillator (product, f=137 minus mix=0): %.3f ms; of which double precision (product minus osc_f32): %.3f ms; demapper (product minus nodemap): %.3f ms"
          % (p - rows[("product", "mix=0")], p - rows[("osc_f32", "mix=1 f=137 again")], p - rows[("nodemap", "mix=1 f=137 again")]))
except KeyError as e:
    print("incomplete:", e)
