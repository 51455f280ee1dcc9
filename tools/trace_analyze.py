"""Not a test: summarises a rocprofv3 kernel-trace CSV of bench.py: per-kernel durations of the sync chain, and for one
steady-state step which decode kernel was running while each chain kernel waited / ran."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(r["Kernel_Name"].split("(")[0].replace("dabphy::", "").replace("void ", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "")) for r in rows if "dabphy" in r["Kernel_Name"]]
ks.sort(key=lambda x: x[1])
CH = ("k_sync_find", "k_cp_products", "k_sync_finish", "k_acquire", "k_sync_chain")
chain = [k for k in ks if k[0] in CH]
heavy = [k for k in ks if k[0] not in CH]
# steady state: take the last full step = between the last two k_demod starts
dem = [k for k in heavy if k[0].startswith("k_demod")]
t_a, t_b = dem[-2][1], dem[-1][1]
print("step window %.2f ms" % ((t_b - t_a) / 1e6))
print("-- decode kernels in the window")
for k in heavy:
    if t_a <= k[1] < t_b:
        print("  %-18s %8.1f -> %8.1f us (%.1f)" % (k[0], (k[1] - t_a) / 1e3, (k[2] - t_a) / 1e3, (k[2] - k[1]) / 1e3))
def phase_at(t):
    for k in heavy:
        if k[1] <= t < k[2]:
            return k[0]
    return "idle"
dur = collections.defaultdict(lambda: collections.defaultdict(list)); gaps = collections.defaultdict(list)
prev_end = None
for k in chain:
    if not (t_a <= k[1] < t_b):
        prev_end = k[2]; continue
    ph = phase_at(k[1])
    dur[k[0]][ph].append((k[2] - k[1]) / 1e3)
    if prev_end is not None:
        gaps[ph].append((k[1] - prev_end) / 1e3)
    prev_end = k[2]
print("-- chain kernel durations [us] by the decode kernel running at their start")
for n, d in dur.items():
    for ph, v in sorted(d.items()):
        v2 = sorted(v); print("  %-14s during %-16s n %3d  median %7.1f  max %7.1f  sum %8.1f" % (n, ph, len(v), v2[len(v2) // 2], v2[-1], sum(v)))
print("-- gaps before chain kernels [us] by phase")
for ph, v in sorted(gaps.items()):
    v2 = sorted(v); print("  %-16s n %3d  median %7.1f  max %7.1f  sum %8.1f" % (ph, len(v), v2[len(v2) // 2], v2[-1], sum(v)))
