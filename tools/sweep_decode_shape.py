"""Not a test: the three Viterbi decoders against each other over batch sizes (the canonical ensemble, 18 x 64 kbit/s + FIC = 76 code words per
frame): per (ensembles, frames per call) the whole dabphy_process call (host clock, steady state) and the decode launch alone
(dabphy_time_fused_msc) with dabphy_config.decode_shape = 1 (one LANE per code word, k_viterbi_fused), = 2 (state-parallel, two code words per
wavefront, k_viterbi_sp2 + its traceback pass k_traceback_sp2) and = 3 (round 4's state-parallel kernel, one code word per wavefront, k_viterbi_sp).  Where the state-parallel kernel stops winning is what dabphy's default (decode_shape = 0) switches at.
  python tools/sweep_decode_shape.py            table (profiles/r04_viterbi_state_parallel.txt)
  python tools/sweep_decode_shape.py --json     one JSON line with the single-ensemble rows and the 32 x 16 row (bench.py's extras.short_batches)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package, PKG_DIR  # noqa: E402

load_package()
import torch  # noqa: E402
from welle_io_amd import capi, workload  # noqa: E402

lib = os.environ.get("DABPHY_LIB", os.path.join(PKG_DIR, "libdabphy_hip.so"))
as_json = "--json" in sys.argv
GEOM = [(1, 1), (1, 4), (1, 8), (1, 16), (32, 16)] if as_json else [(1, 1), (1, 4), (1, 8), (1, 16), (4, 4), (8, 8), (16, 8), (16, 16), (32, 16), (48, 16), (64, 16), (128, 16)]
base = workload.make_base_streams(2, workload.REC_FRAMES, seed0=0)
rows = []
for B, F in GEOM:
    iq, cfo, base_np, txs = workload.make_batch(B, base=base, device="cuda")
    rec = {"ensembles": B, "frames_per_call": F, "code_words": B * F * 76}
    # (--json, bench.py's extras.short_batches: the lane-per-code-word kernel against what the library picks by itself for these sizes,
    # decode_shape = 0 -- k_viterbi_sp up to 1024 code words per call, k_viterbi_sp2 + k_traceback_sp2 for the 16-frame row; the table compares all three explicitly)
    for shape, name in ((1, "lane_per_codeword"), (0 if as_json else 2, "state_parallel")) + (() if as_json else ((3, "state_parallel_r4"),)):
        dev = workload.open_receiver(capi, lib, iq, F, txs[0].subchs, pipeline_sync=0, profiling=True, decode_shape=shape)
        dev.set_auto_superframes(False)
        for _ in range(4):
            dev.process(F)
        torch.cuda.synchronize()
        n = max(5, 200 // (B * F)); t0 = time.perf_counter(); acc = 0.0
        for _ in range(n):
            dev.process(F); fib, ok = dev.fibs_host(); acc += dev.stage_times()["msc_viterbi"]
        dt = (time.perf_counter() - t0) / n
        assert np.asarray(ok).all()
        rec[name] = {"ms_per_call": dt * 1e3, "x_real_time": B * F * 0.096 / dt, "decode_ms_in_call": acc / n, "decode_ms_alone": dev.time_fused_msc(5)}
        dev.close()
    rows.append(rec)
    if not as_json:
        a, b, c = rec["lane_per_codeword"], rec["state_parallel"], rec["state_parallel_r4"]
        print("%4d ensembles x %2d frames (%6d code words): lane-per-code-word %7.3f ms per call, decode alone %7.3f ms | state-parallel (2 code words per wave) %7.3f ms per call, decode alone %7.3f ms | round 4's (1 per wave) %7.3f / %7.3f | %s"
              % (B, F, rec["code_words"], a["ms_per_call"], a["decode_ms_alone"], b["ms_per_call"], b["decode_ms_alone"], c["ms_per_call"], c["decode_ms_alone"], "state-parallel wins" if b["decode_ms_alone"] < a["decode_ms_alone"] else "lane-per-code-word wins"), flush=True)
if as_json:
    print(json.dumps({"what": "one ensemble (last row: 32 ensembles), F frames per dabphy_process call (serial synchroniser, FIBs copied out): per-call latency and x real-time with the lane-per-code-word kernel forced (decode_shape 1) and with the library's own choice (decode_shape 0: state-parallel -- k_viterbi_sp, above 1024 code words k_viterbi_sp2 + k_traceback_sp2)", "rows": rows}))
