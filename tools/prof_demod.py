"""Not a test: runs only the demod kernel (B x F frame slots) so that rocprofv3 --pmc passes stay short."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import GPU_LIB  # noqa: E402
import parity_cases as P  # noqa: E402
from welle_io_amd import capi, synth  # noqa: E402

B = int(os.environ.get("PROBE_B", "256")); F = int(os.environ.get("PROBE_F", "32"))
x = synth.make_stream(5, snr_db=20, seed=1)
frames = P.cut_frames(x, 4)
d = capi.DabPhy(lib_path=GPU_LIB, demod_chunk=int(os.environ.get("PROBE_CHUNK", "15")))
print(d.time_demod(frames, B, F, mix=int(os.environ.get("PROBE_MIX", "1")), f_hz=0, iters=2))
d.close()
