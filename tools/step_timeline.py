"""Not a test: prints the kernels of the last complete step of a rocprofv3 --kernel-trace run of bench.py (start, duration, stream),
from gpurun_out/kt/kt_kernel_trace.csv.  usage: python tools/step_timeline.py [csv [n]]  (n: the step that starts at the n-th demod launch instead of the last complete one)"""
import csv
import sys

f = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/kt/kt_kernel_trace.csv"
rows = [r for r in csv.DictReader(open(f)) if "dabphy" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_demod" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) - 2
start = idx[k]; t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:idx[k + 1] + 1]:
    n = r["Kernel_Name"].split("(")[0].replace("dabphy::", "").replace("void ", "")
    print("%-28s start %8.3f ms  dur %7.3f ms  end %8.3f ms  stream %s" % (n, (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6,
                                                                       (int(r["End_Timestamp"]) - t0) / 1e6, r.get("Stream_Id", "?")))
