"""k_viterbi time vs number of 64-codeword groups per SIMD (occupancy rounds / tail).  GPU box: python tools/probe_vit_tail.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: F401,E402
from conftest import GPU_LIB  # noqa: E402
from welle_io_amd import capi  # noqa: E402

d = capi.DabPhy(lib_path=os.environ.get("DABPHY_LIB", GPU_LIB), n_ensembles=1, max_frames=1)
for groups in [int(a) for a in sys.argv[1:]] or (1024, 2048, 3072, 4096, 4608, 5120, 5440, 5760, 6144, 7168, 8192, 10240, 11520):
    r = d.time_viterbi(1536, groups * 64, iters=5)
    print("groups %5d (%.3f per SIMD): gather %.3f ms  viterbi %.3f ms  -> %.3f us per group-per-SIMD" % (groups, groups / 1024, r[0], r[1], 1e3 * r[1] / (groups / 1024)))
d.close()
