import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import GPU_LIB, EMU_LIB
import parity_cases as P, refapi as R
from welle_io_amd import capi, synth
lib = EMU_LIB if os.environ.get("EMU") else GPU_LIB
x, tx = synth.make_stream(9, snr_db=None, return_tx=True, seed=3)
subs = [tx.subchs[0]]
o = R.orc_receiver_run(x, subchs=subs, want_soft=True)
print("oracle frames", o["n_frames"], "start", o["start_index"].tolist(), "pos", o["frame_pos"].tolist(), "corr", o["corr"].tolist())
d = capi.DabPhy(lib_path=lib, max_frames=4)
d.stream_upload(x[None, :]); d.set_subchannels([(s.subch_id, s.start_cu, s.size_cu, d.protection_eep(s.bitrate, s.profile_b, s.level)) for s in subs])
for it in range(3):
    d.process(4)
    info = d.frame_info()
    print("dev", info[0]["pos"].tolist(), info[0]["start_index"].tolist(), info[0]["valid"].tolist(), info[0]["fine"].tolist(), info[0]["coarse"].tolist())
    fb, ok = d.fibs(); print(" ok", ok[0].sum(axis=1).tolist())
