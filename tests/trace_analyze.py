"""Not a test: summarises a rocprofv3 kernel trace CSV (durations and queue gaps of the sync-chain kernels)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(r["Kernel_Name"].split("(")[0].replace("dabphy::", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "")) for r in rows if "dabphy" in r["Kernel_Name"]]
ks.sort(key=lambda x: x[1])
t0 = ks[len(ks) // 2][1]
chain = [k for k in ks if k[0] in ("k_sync_frame", "k_cp_products", "k_acquire")]
dur = collections.defaultdict(list); gap = []
for i, k in enumerate(chain):
    dur[k[0]].append((k[2] - k[1]) / 1e3)
    if i: gap.append((k[1] - chain[i - 1][2]) / 1e3)
for n, v in dur.items():
    v2 = sorted(v); print(n, "n", len(v), "median %.1f us  p90 %.1f  max %.1f" % (v2[len(v2) // 2], v2[int(len(v2) * 0.9)], v2[-1]))
g2 = sorted(gap); print("gaps between chain kernels: median %.1f us p90 %.1f max %.1f sum %.1f ms" % (g2[len(g2) // 2], g2[int(len(g2) * 0.9)], g2[-1], sum(gap) / 1e3))
# timeline of the last batch
last = ks[-60:]
base = last[0][1]
for k in last:
    print("%-16s %9.1f -> %9.1f us  (%.1f)  q=%s" % (k[0], (k[1] - base) / 1e3, (k[2] - base) / 1e3, (k[2] - k[1]) / 1e3, k[3]))
