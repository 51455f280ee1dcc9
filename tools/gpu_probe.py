"""Not a test: stage timings on the GPU box (kernel-level HIP events), used while tuning."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import GPU_LIB  # noqa: E402
import parity_cases as P  # noqa: E402
from welle_io_amd import capi, synth  # noqa: E402

B = int(os.environ.get("PROBE_B", "256")); F = int(os.environ.get("PROBE_F", "20"))
x = synth.make_stream(5, snr_db=20, seed=1)
frames = P.cut_frames(x, 4)
res = {}
for chunk in (15, 25):
    d = capi.DabPhy(lib_path=GPU_LIB, demod_chunk=chunk)
    for mix, f_hz in ((0, 0), (1, 0), (1, 137)):
        ms = d.time_demod(frames, B, F, mix=mix, f_hz=f_hz, iters=5)
        nfr = B * F
        res["demod_chunk%d_mix%d_f%d" % (chunk, mix, f_hz)] = dict(ms=ms, frames_per_s=nfr / ms * 1e3, xRT=nfr * 0.096 / (ms * 1e-3),
                                                                 GBps_alg=nfr * (76 * 16384 + 75 * 3072) / ms / 1e6)
    d.close()
d = capi.DabPhy(lib_path=GPU_LIB)
for nbits, ncw in ((768, B * F * 4), (1536, B * F * 72)):
    g, v = d.time_viterbi(nbits, ncw, iters=3)
    res["viterbi_%d_x%d" % (nbits, ncw)] = dict(ms_gather=g, ms_decode=v, cw_steps_per_s=ncw * (nbits + 6) / v * 1e3)
print(json.dumps({k: {kk: round(vv, 3) for kk, vv in v.items()} for k, v in res.items()}))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/probe.json", "w"), indent=1)
