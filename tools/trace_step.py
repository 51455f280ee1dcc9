"""Timeline of the last full pipelined step in a rocprofv3 kernel trace (gpurun_out/<dir>/<prefix>_kernel_trace.csv):
start / duration (us, relative to the start of k_demod) of every product kernel, chain kernels as start:duration lists."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("dabphy::", "").replace("void ", "")) for r in rows)
dem = [i for i, e in enumerate(ev) if e[2].startswith("k_demod")]
i0 = dem[-2]; t0 = ev[i0][0]; end = ev[dem[-1]][0]
chain = {}
for s, e, n in ev:
    if t0 <= s < end:
        if n in ("k_sync_find", "k_cp_products", "k_sync_finish"):
            chain.setdefault(n, []).append(((s - t0) / 1e3, (e - s) / 1e3))
        elif not n.startswith("at::") and not n.startswith("__amd"):
            print("%-18s start %8.1f us  dur %8.1f us" % (n, (s - t0) / 1e3, (e - s) / 1e3))
for n, l in chain.items():
    print(n, " ".join("%.0f:%.0f" % (a, b) for a, b in l))
print("step (demod start to next demod start): %.1f us" % ((end - t0) / 1e3))
