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
def run_build(lib, nf):
    """the reference's own RadioReceiver (all its threads) behind the recording harness, from the build `lib`, over nf frames"""
    x, tx = synth.make_stream(nf, snr_db=20, cfo_hz=40, delay=100, seed=3, return_tx=True)
    t = time.time()
    a = R.receiver_run(x, subchs=[tx.subchs[2], tx.subchs[11]], lib=lib)
    return time.time() - t, a


def level2():
    """INTEGRATION.md level 2 (BASELINE config 2): the reference backend with ONE source file replaced by a seam binding, next to the
    unmodified build on the same host: ms per 96 ms frame (slope between 30 and 90 frames), one ensemble, two services selected"""
    out = {}
    for name, lib, what in (("reference", None, "the unmodified reference backend (CPU)"),
                            ("l2a", R.level2_lib("a", "hip"), "ofdm-decoder.cpp -> seams/ofdm_decoder_seam.cpp: FFT + DQPSK demap + frequency de-interleaver on the device, Viterbi on the CPU"),
                            ("l2b", R.level2_lib("b", "hip"), "viterbi.cpp -> seams/viterbi_seam.cpp: every Viterbi::deconvolve on the device (one code word per call), the rest on the CPU")):
        if lib is not None and not os.path.exists(lib):
            out[name] = {"error": "%s not built" % os.path.basename(lib)}
            continue
        try:
            run_build(lib, 6)
            ta, _ = run_build(lib, 30)
            tb, a = run_build(lib, 90)
            ms = (tb - ta) / 60 * 1e3
            out[name] = {"what": what, "ms_per_frame": ms, "x_realtime": 96.0 / ms, "fib_crc_ok": int(a["fib"][:, 0].sum()), "fibs": int(len(a["fib"]))}
        except Exception as ex:
            out[name] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    return out


if "--json" in sys.argv:
    ms = (t2 - t1) / 80 * 1e3
    print(json.dumps({"level2": level2() if R.have_ref() else {"error": "oracle/_ref not built"},
                      "what": "GpuRadioReceiver (drop-in for RadioReceiver) over one synthetic ensemble: dabphy_process(1) per 96 ms frame, every getter copied to the host, the reference's FIBProcessor fed on the host (BASELINE configs 2-3)",
                      "ms_per_frame": ms, "x_realtime": 96.0 / ms, "frames": 120, "fib_crc_ok": int(b["fib"][:, 0].sum()), "fibs": int(len(b["fib"]))}))
    sys.exit(0)
print("%.2f ms per 96 ms frame in steady state (slope between 40 and 120 frames; %.0f ms fixed: handle, tables, acquisition)  FIBs ok %d of %d"
      % ((t2 - t1) / 80 * 1e3, (t1 - 40 * (t2 - t1) / 80) * 1e3, int(b["fib"][:, 0].sum()), len(b["fib"])))
