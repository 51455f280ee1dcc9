"""Not a test: random MULTIPLEXES through the streaming receiver on the GPU box -- what the generalised decoders of round 4 are for.  Every
stream draws its own sub-channel layout (3 ... 14 sub-channels; EEP profile A at 8 ... 192 kbit/s and B at 32 ... 192 kbit/s, levels 1-4;
UEP rows of the table; code words of 192 ... 4608 bits in as many protection classes as come up), batch depth (1 ... 20 frames per call: all
three builds of the fused kernel's window ring), number of ensembles (2 ... 6) and Viterbi kernel (dabphy_config.decode_shape 0 = the default's
choice, 1 = lane per code word, 2 = k_viterbi_sp2 + k_traceback_sp2, 3 = k_viterbi_sp), and compares FIBs, CRC flags and the MSC bytes of EVERY sub-channel of every ensemble with the oracle
(tests/parity_cases.check_mixed_ensemble): no tolerance.
python tools/sweep_multiplex.py [n_streams] [seed]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: F401,E402
from conftest import GPU_LIB  # noqa: E402
import parity_cases as P  # noqa: E402
from welle_io_amd import capi, synth, workload  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lib_path = os.environ.get("DABPHY_LIB", GPU_LIB)
lib = capi.load_library(lib_path)
uep_rows = []
for idx in range(64):
    size = C.c_int(0); lvl = C.c_int(0); br = C.c_int(0)
    if lib.dabphy_uep_table_entry(idx, C.byref(size), C.byref(lvl), C.byref(br)) == 0 and br.value > 0 and br.value <= 192:
        uep_rows.append((br.value, lvl.value))


def random_layout():
    subchs = []; cu = 0
    want = int(rng.randint(3, 15))
    for sid in range(1, want + 1):
        for _ in range(8):                                 # a few draws until one fits the 864 capacity units
            kind = rng.rand()
            if kind < 0.2:
                br, lvl = uep_rows[int(rng.randint(len(uep_rows)))]
                sc = workload.uep_subchannel(lib, sid, cu, br, lvl, dabplus=False)
            elif kind < 0.45:
                sc = synth.SubchannelCfg(sid, cu, int(rng.choice([32, 64, 96, 128, 192])), True, int(rng.randint(1, 5)), dabplus=False)
            else:
                sc = synth.SubchannelCfg(sid, cu, int(rng.choice([8, 16, 24, 32, 40, 48, 56, 64, 72, 80, 96, 112, 128, 160, 192])), False, int(rng.randint(1, 5)), dabplus=False)
            if cu + sc.size_cu <= 864:
                subchs.append(sc); cu += sc.size_cu
                break
    return subchs


tot_frames = 0
for it in range(n):
    subchs = random_layout()
    F = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 12, 16, 20])); B = int(rng.choice([2, 3, 6])); shape = int(rng.choice([0, 1, 2, 3]))
    nf = max(7, 2 * F + 6); snr = float(rng.choice([9, 12, 16, 22])); seed = int(rng.randint(1 << 30))
    P.check_mixed_ensemble(lambda **kw: capi.DabPhy(lib_path=lib_path, decode_shape=shape, **kw), F=F, nf=nf, snr_db=snr, seed=seed, B=B, subchs=subchs, expect_fused=True)
    classes = len({(s.bitrate, s.profile_b, s.level, s.uep is not None) for s in subchs})
    tot_frames += nf
    print("stream %3d  %2d sub-channels in %2d classes (%s)  %2d frames per call x %d ensembles  decode_shape %d  snr %2.0f dB  frames %d: equal"
          % (it, len(subchs), classes, " ".join("%d%s%d" % (s.bitrate, "U" if s.uep is not None else ("B" if s.profile_b else "A"), s.level) for s in subchs), F, B, shape, snr, nf), flush=True)
print("streams %d  frames %d  mismatches 0" % (n, tot_frames))
