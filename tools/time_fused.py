"""Not a test: times the fused decode launch (k_viterbi_fused: every MSC class + the FIC) ALONE on the benchmark batch (dabphy_time_fused_msc:
the launch of the last batch re-run with nothing else on the device) for A/B runs on the GPU box.  usage: python tools/time_fused.py [B] [F] ; DABPHY_LIB selects the library."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package, PKG_DIR  # noqa: E402

load_package()
import torch  # noqa: E402
from welle_io_amd import capi, synth, workload  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
F = int(sys.argv[2]) if len(sys.argv) > 2 else 32
lib = os.environ.get("DABPHY_LIB", os.path.join(PKG_DIR, "libdabphy_hip.so"))
rec = workload.rec_frames_for(F)
iq, cfo, base, txs = workload.make_batch(B, rec_frames=rec)
dev = workload.open_receiver(capi, lib, iq, F, txs[0].subchs)
for _ in range(3):
    dev.process(F)
torch.cuda.synchronize()
ms = dev.time_fused_msc(5)
groups = B * 18 * 4 * F // 64
steps = 1542
print("k_viterbi_fused alone (%s): %.3f ms per launch (%d x %d, %d MSC groups + the FIC class if fused); %.0f cycles per MSC trellis step and SIMD at 2.4 GHz" % (os.path.basename(lib), ms, B, F, groups, ms * 1e-3 * 2.4e9 / (groups * steps / 1024.0)))
dev.close()
