"""(--json: one JSON line, what bench.py embeds as `facade`)  Latency of the drop-in receiver (GpuRadioReceiver: one frame per dabphy_process, every getter copied to the host, the reference's
FIBProcessor on the host) over a synthetic stream on the GPU box: python tools/bench_facade.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: F401,E402
import refapi as R  # noqa: E402
from welle_io_amd import synth  # noqa: E402

def run(nf):
    x = synth.make_stream(nf, snr_db=20, cfo_hz=40, delay=100, seed=3)
    t = time.time()
    b = R.gpu_receiver_run(x, lib=R.GPU_HIP_SO)
    return time.time() - t, b


import json

if not os.path.exists(R.GPU_HIP_SO):
    print(json.dumps({"error": "oracle/_ref/libwelle_gpu_hip.so not built (it links the reference's FIBProcessor: needs /root/reference at build time)"}))
    sys.exit(0)
run(6)                                                        # warm-up: library, tables
t1, _ = run(40)
t2, b = run(120)
if "--json" in sys.argv:
    ms = (t2 - t1) / 80 * 1e3
    print(json.dumps({"what": "GpuRadioReceiver (drop-in for RadioReceiver) over one synthetic ensemble: dabphy_process(1) per 96 ms frame, every getter copied to the host, the reference's FIBProcessor fed on the host (BASELINE configs 2-3)",
                      "ms_per_frame": ms, "x_realtime": 96.0 / ms, "frames": 120, "fib_crc_ok": int(b["fib"][:, 0].sum()), "fibs": int(len(b["fib"]))}))
    sys.exit(0)
print("%.2f ms per 96 ms frame in steady state (slope between 40 and 120 frames; %.0f ms fixed: handle, tables, acquisition)  FIBs ok %d of %d"
      % ((t2 - t1) / 80 * 1e3, (t1 - 40 * (t2 - t1) / 80) * 1e3, int(b["fib"][:, 0].sum()), len(b["fib"])))
