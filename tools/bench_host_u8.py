"""Not a test, not the headline metric: end-to-end rate when the samples start in HOST memory as u8 IQ (what a file or an
RTL-SDR delivers), i.e. including PCIe and the on-device conversion.  One step = B ensembles x F frames appended to the
library's ring (dabphy_stream_write_raw from page-locked memory) and decoded.  Prints one JSON line; DESIGN.md quotes it."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package, PKG_DIR  # noqa: E402

B = int(os.environ.get("HOSTU8_B", "64")); F = int(os.environ.get("HOSTU8_F", "20")); STEPS = int(os.environ.get("HOSTU8_STEPS", "4"))
T_F = 196608
load_package()
from welle_io_amd import capi, synth  # noqa: E402

tx = synth.EnsembleTx(eid=0x1000, seed=1, payload_fn=synth.dabplus_payload_fn(80 * -(-4 * F // 80), 1))      # (payload period: whole superframes and interleaver periods; a block of F frames is then re-sent every step -- the time interleaver sees a splice per step, the rate does not care)
for _ in range(F):
    tx.next_frame()
x = np.concatenate([tx.next_frame() for _ in range(F)]).astype(np.complex64)
u8 = synth.to_u8(x).reshape(-1, 2)
dev = capi.DabPhy(n_ensembles=B, max_frames=F, lib_path=os.path.join(PKG_DIR, "libdabphy_hip.so"), want_constellation=False, want_impulse_response=False)
pin = dev.host_alloc((B, F * T_F, 2), np.uint8)
pin[:] = u8[None]
dev.stream_open((2 * F + 4) * T_F)
subs = tx.subchs
dev.set_subchannels([(s.subch_id, s.start_cu, s.size_cu, dev.protection_eep(s.bitrate, s.profile_b, s.level)) for s in subs])
ASYNC = os.environ.get("HOSTU8_ASYNC", "1") != "0"
pin2 = dev.host_alloc((B, F * T_F, 2), np.uint8); pin2[:] = pin
bufs = [pin, pin2]; turn = [0]
if ASYNC:
    dev.stream_write_raw_async(bufs[0], "u8"); turn[0] = 1       # batch 0 is on its way before the first process()
def step():
    if ASYNC:                                           # batch k+1 crosses PCIe while batch k is decoded
        dev.stream_commit()
        dev.stream_write_raw_async(bufs[turn[0] & 1], "u8"); turn[0] += 1
        dev.process(F)
    else:
        dev.stream_write_raw(pin, "u8")                 # one contiguous page-locked block: B x F frames of u8 IQ, then decode
        dev.process(F)
    return dev.fibs()
step()
fib, ok = step()
t0 = time.perf_counter()
for _ in range(STEPS):
    fib, ok = step()
dt = (time.perf_counter() - t0) / STEPS
print(json.dumps({"what": "host u8 -> FIBs/MSC bytes, PCIe and conversion included (%s)" % ("copy of batch k+1 overlapped with the decode of batch k" if ASYNC else "serial: copy, then decode"), "ensembles": B, "frames_per_step": F,
                  "ms_per_step": dt * 1e3, "x_real_time": B * F * 0.096 / dt, "host_GBps": B * F * T_F * 2 / dt / 1e9, "fib_crc_ok": float(ok.mean())}))
dev.close()
