"""Not a test: ONE of bench.py's channel legs on its own (extras.drift / extras.low_snr: the headline's geometry on drifting sample clocks /
at 6-10 dB), so that a profiler can wrap exactly that workload (tools/make_profiles.sh: rocprofv3 --kernel-trace --stats -> profiles/
rNN_drift_kernel_stats.csv).  usage: python tools/bench_channel.py drift|low_snr [B] [F] [steps]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package, PKG_DIR  # noqa: E402

load_package()
import torch  # noqa: E402
import bench  # noqa: E402
from welle_io_amd import capi, workload  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "drift"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
F = int(sys.argv[3]) if len(sys.argv) > 3 else 32
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
torch.cuda.init()
lib = os.environ.get("DABPHY_LIB", os.path.join(PKG_DIR, "libdabphy_hip.so"))
print(json.dumps(bench.channel_leg(capi, workload, torch, lib, B, F, steps, 0, int(os.environ.get("DABPHY_PIPELINE", "1")), kind)), flush=True)
