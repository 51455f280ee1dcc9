"""Not a test: driver for a rocprofv3 --kernel-trace of ONE batch geometry in latency mode (serial synchroniser, FIBs copied out after every
call), to see what a dabphy_process call of a small or medium batch is made of besides its decode launch:
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -o kt -- python tools/trace_call.py <ensembles> <frames per call> <decode_shape> [calls]
  python tools/step_timeline.py gpurun_out/kt/kt_kernel_trace.csv"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package, PKG_DIR  # noqa: E402

load_package()
import torch  # noqa: E402
from welle_io_amd import capi, workload  # noqa: E402

B, F, shape = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 12
lib = os.environ.get("DABPHY_LIB", os.path.join(PKG_DIR, "libdabphy_hip.so"))
base = workload.make_base_streams(2, workload.REC_FRAMES, seed0=0)
iq, cfo, base_np, txs = workload.make_batch(B, base=base, device="cuda")
dev = workload.open_receiver(capi, lib, iq, F, txs[0].subchs, pipeline_sync=0, profiling=False, decode_shape=shape)
dev.set_auto_superframes(False)
for _ in range(4):
    dev.process(F)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(calls):
    dev.process(F); dev.fibs_host()
print("%d x %d decode_shape %d: %.3f ms per call (host clock)" % (B, F, shape, (time.perf_counter() - t0) / calls * 1e3))
dev.close()
