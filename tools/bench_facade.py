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
def run_build(lib, nf, n_services=2):
    """the reference's own RadioReceiver (all its threads) behind the recording harness, from the build `lib`, over nf frames;
    -> (wall seconds, CPU seconds of the whole process: every thread of the receiver, user + system, result)"""
    import resource
    x, tx = synth.make_stream(nf, snr_db=20, cfo_hz=40, delay=100, seed=3, return_tx=True)
    subs = [tx.subchs[2], tx.subchs[11]] if n_services == 2 else list(tx.subchs[:n_services])
    r0 = resource.getrusage(resource.RUSAGE_SELF); t = time.time()
    a = R.receiver_run(x, subchs=subs, lib=lib)
    dt = time.time() - t; r1 = resource.getrusage(resource.RUSAGE_SELF)
    return dt, (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime), a


def seam_stats(lib):
    """device calls / code words of the Viterbi seam's shared decoder (welle.io_amd/host/seams/viterbi_seam.cpp) so far in this process"""
    import ctypes as C
    try:
        L = C.CDLL(lib); a = C.c_ulonglong(0); b = C.c_ulonglong(0)
        L.dabphy_seam_viterbi_stats(C.byref(a), C.byref(b))
        return a.value, b.value
    except Exception:
        return None


QUICK = os.environ.get("DABPHY_BENCH_QUICK") == "1"      # tests only: shorter streams (the test checks that the figures exist, not what they are)
N_A, N_B = (12, 36) if QUICK else (30, 90)


def level2():
    """INTEGRATION.md level 2 (BASELINE config 2): the reference backend with ONE source file replaced by a seam binding, next to the
    unmodified build on the same host, one ensemble: wall ms and CPU ms (all threads of the receiver) per 96 ms frame (slopes between 30
    and 90 frames -- 12 and 36 with DABPHY_BENCH_QUICK=1 --; the harness waits 0.5 s for the decoders to go quiet: that wait is in neither slope), with two services selected and
    with all 18 -- where the channel decoders are what the host's cores do (viterbi.cpp: 24 of the reference's 30 CPU ms per frame)"""
    out = {}
    for name, lib, what in (("reference", None, "the unmodified reference backend (CPU)"),
                            ("l2a", R.level2_lib("a", "hip"), "ofdm-decoder.cpp -> seams/ofdm_decoder_seam.cpp: FFT + DQPSK demap + frequency de-interleaver on the device, Viterbi on the CPU"),
                            ("l2b", R.level2_lib("b", "hip"), "viterbi.cpp -> seams/viterbi_seam.cpp: every Viterbi::deconvolve on the device through ONE shared handle; concurrent calls (the sub-channels of a CIF) are combined into one batch per code word length, the rest on the CPU")):
        if lib is not None and not os.path.exists(lib):
            out[name] = {"error": "%s not built" % os.path.basename(lib)}
            continue
        try:
            run_build(lib, 6)
            ta, ca, _ = run_build(lib, N_A)
            tb, cb, a = run_build(lib, N_B)
            ms = (tb - ta) / (N_B - N_A) * 1e3
            out[name] = {"what": what, "ms_per_frame": ms, "cpu_ms_per_frame": (cb - ca) / (N_B - N_A) * 1e3, "x_realtime": 96.0 / ms, "fib_crc_ok": int(a["fib"][:, 0].sum()), "fibs": int(len(a["fib"]))}
            if name != "l2a":
                s0 = seam_stats(lib) if name == "l2b" else None
                ta, ca, _ = run_build(lib, N_A, 18)
                tb, cb, a = run_build(lib, N_B, 18)
                out[name]["all_18_services"] = {"ms_per_frame": (tb - ta) / (N_B - N_A) * 1e3, "cpu_ms_per_frame": (cb - ca) / (N_B - N_A) * 1e3}
                s1 = seam_stats(lib) if name == "l2b" else None
                if s0 and s1 and s1[0] > s0[0]:
                    out[name]["all_18_services"]["code_words_per_device_call"] = (s1[1] - s0[1]) / (s1[0] - s0[0])
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
