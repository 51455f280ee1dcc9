"""Not a test: copies the rocprofv3 summaries that tools/make_profiles.sh left under gpurun_out/prof/ into profiles/
(tracked) and derives profiles/demod_hbm_traffic.json, the per-launch HBM traffic bench.py reports as roofline.traffic."""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package, PKG_DIR  # noqa: E402
load_package()
from welle_io_amd import buildid  # noqa: E402
# the build these counters belong to (bench.py reports them only while the hashes match the library it runs)
BUILD = {"src_sha256": buildid.source_sha256(), "lib_sha256": buildid.file_sha256(os.path.join(PKG_DIR, "libdabphy_hip.so"))}
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r02"
B, F = int(os.environ.get("PROF_B", "256")), int(os.environ.get("PROF_F", "32"))

shutil.copy(os.path.join(SRC, "stats_kernel_stats.csv"), os.path.join(DST, TAG + "_bench_kernel_stats.csv"))
if os.path.exists(os.path.join(SRC, "drift_kernel_stats.csv")):          # the drift leg alone under the kernel tracer (tools/bench_channel.py)
    shutil.copy(os.path.join(SRC, "drift_kernel_stats.csv"), os.path.join(DST, TAG + "_drift_kernel_stats.csv"))


def rows(name):
    return list(csv.DictReader(open(os.path.join(SRC, name))))


def per_kernel(name, counter):
    acc = {}
    for r in rows(name):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0]
        acc.setdefault(k, []).append(float(r["Counter_Value"]))
    return acc


def dump(name, out, counters):
    rr = [r for r in rows(name) if r["Counter_Name"] in counters]
    keep = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
            "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
    with open(os.path.join(DST, out), "w", newline="") as f:
        w = csv.DictWriter(f, keep); w.writeheader()
        for r in rr:
            w.writerow({k: r.get(k, "") for k in keep})


dump("pmc_fetch_counter_collection.csv", TAG + "_pmc_fetch_size.csv", {"FETCH_SIZE"})
dump("pmc_write_counter_collection.csv", TAG + "_pmc_write_size.csv", {"WRITE_SIZE"})
sq = {"SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_BUSY_CYCLES",
      "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VMEM", "SQ_WAVES", "GRBM_GUI_ACTIVE"}
with open(os.path.join(DST, TAG + "_pmc_sq_demod.csv"), "w", newline="") as f:
    w = None
    for name in ("pmc_sq1_counter_collection.csv", "pmc_sq2_counter_collection.csv"):
        for r in rows(name):
            if r["Counter_Name"] in sq:
                if w is None:
                    w = csv.DictWriter(f, ["Dispatch_Id", "Kernel_Name", "VGPR_Count", "LDS_Block_Size", "Counter_Name", "Counter_Value"]); w.writeheader()
                w.writerow({k: r.get(k, "") for k in w.fieldnames})

fetch = per_kernel("pmc_fetch_counter_collection.csv", "FETCH_SIZE")
write = per_kernel("pmc_write_counter_collection.csv", "WRITE_SIZE")
kd = [k for k in fetch if "k_demod" in k][0]
# steady-state launches only (the first launch of a run demodulates fewer frames: acquisition)
fk = sorted(fetch[kd])[len(fetch[kd]) // 2]; wk = sorted(write[kd])[len(write[kd]) // 2]
alg = B * F * (76 * 2048 * 8 + 75 * 3072)
hbm = int(fk * 1024 * 2 + wk * 1024)
json.dump({
    "kernel": "dabphy::k_demod", "ensembles": B, "frames": F,
    "fetch_size_kb_raw": fk, "write_size_kb_raw": wk,
    "correction": "FETCH_SIZE doubled (gfx950 rocprofv3 counts 128-B requests of wide coalesced streams at 64 B, MI355X_MICROARCH.md 'HBM'); WRITE_SIZE as reported",
    "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg,
    "command": "tools/make_profiles.sh: rocprofv3 --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) --kernel-include-regex ... -- python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-alt-schedule",
    "source": "profiles/%s_pmc_fetch_size.csv, profiles/%s_pmc_write_size.csv (median launch)" % (TAG, TAG),
    "traffic_over_algorithmic": hbm / alg, **BUILD,
}, open(os.path.join(DST, "demod_hbm_traffic.json"), "w"), indent=1)
print(open(os.path.join(DST, "demod_hbm_traffic.json")).read())
for k in sorted(fetch):
    print("%-60s fetch %10.0f KB  write %10.0f KB" % (k[:60], sorted(fetch[k])[len(fetch[k]) // 2], sorted(write.get(k, [0]))[len(write.get(k, [0])) // 2]))


# ---- Viterbi stage: instruction counts and HBM traffic of the MSC launch (grid = B*F*72/64 one-wave work-groups), for bench.py's
# roofline_viterbi block; the ubench's issue costs
vit_grid = B * F * 72            # work-items of the MSC launch (64 per group) on the two-kernel path


def is_msc_launch(r):
    """the fused MSC decode (any grid: a work-group walks several 64-codeword groups when there are more groups than wave slots), or
    the MSC launch of the two-kernel path (the FIC class uses the same k_viterbi with a smaller grid)"""
    name = r["Kernel_Name"]
    return "k_viterbi_fused" in name or "k_viterbi_msc" in name or ("k_viterbi(" in name and int(r["Grid_Size"]) == vit_grid)


if os.path.exists(os.path.join(SRC, "pmc_vit_counter_collection.csv")):
    shutil.copy(os.path.join(SRC, "pmc_vit_counter_collection.csv"), os.path.join(DST, TAG + "_pmc_sq_viterbi_gather.csv"))
    acc = {}
    for r in rows("pmc_vit_counter_collection.csv"):
        if is_msc_launch(r):
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    med = {k: sorted(v)[len(v) // 2] for k, v in acc.items()}
    fv = [float(r["Counter_Value"]) for r in rows("pmc_fetch_counter_collection.csv") if r["Counter_Name"] == "FETCH_SIZE" and is_msc_launch(r)]
    wv = [float(r["Counter_Value"]) for r in rows("pmc_write_counter_collection.csv") if r["Counter_Name"] == "WRITE_SIZE" and is_msc_launch(r)]
    fg = [float(r["Counter_Value"]) for r in rows("pmc_fetch_counter_collection.csv") if r["Counter_Name"] == "FETCH_SIZE" and "k_msc_gather" in r["Kernel_Name"]]
    wg = [float(r["Counter_Value"]) for r in rows("pmc_write_counter_collection.csv") if r["Counter_Name"] == "WRITE_SIZE" and "k_msc_gather" in r["Kernel_Name"]]
    mid = lambda v: sorted(v)[len(v) // 2] if v else 0.0
    json.dump({
        "kernel": "dabphy::k_viterbi_fused (all MSC classes + the FIC in one launch: gathers fused into the Viterbi kernel)", "ensembles": B, "frames": F,
        "valu_insts_per_launch": med.get("SQ_INSTS_VALU"), "lds_insts_per_launch": med.get("SQ_INSTS_LDS"), "vmem_insts_per_launch": med.get("SQ_INSTS_VMEM"),
        "waves": med.get("SQ_WAVES"), "lds_bank_conflict_cycles": med.get("SQ_LDS_BANK_CONFLICT"), "lds_idx_active_cycles": med.get("SQ_LDS_IDX_ACTIVE"),
        "fetch_size_kb_raw": mid(fv), "write_size_kb_raw": mid(wv),
        "hbm_bytes_per_launch": int(mid(fv) * 1024 * 2 + mid(wv) * 1024),
        "gather_fetch_size_kb_raw": mid(fg), "gather_write_size_kb_raw": mid(wg), "gather_hbm_bytes_per_launch": int(mid(fg) * 1024 * 2 + mid(wg) * 1024),
        "algorithmic_bytes_per_launch": B * F * (72 * (4 * 1542 + 1536 // 8) + 4 * (4 * 774 + 768 // 8)),
        # what is known without counters: the decision array, 8 bytes per trellis step and code word, written once and read back once
        "decision_bytes_written_plus_read": 2 * ((B * F * 72 // 64) * 1542 + (B * F * 4 // 64) * 774) * 512,
        # the guide calibrates the doubling for WIDE coalesced reads (128-byte requests tallied at 64): the decision reads are such reads,
        # the soft-bit windows are 4-byte-per-lane LDS-DMA requests.  If those are tallied at their true 64 bytes: reads = decisions
        # (known) + (raw FETCH - decisions / 2)
        "hbm_bytes_if_narrow_requests_are_tallied_in_full": int(((B * F * 72 // 64) * 1542 + (B * F * 4 // 64) * 774) * 512 + (mid(fv) * 1024 - ((B * F * 72 // 64) * 1542 + (B * F * 4 // 64) * 774) * 512 / 2) + mid(wv) * 1024),
        "correction": "FETCH_SIZE doubled (gfx950, MI355X_MICROARCH.md 'HBM'); WRITE_SIZE as reported; median launch",
        "source": "profiles/%s_pmc_sq_viterbi_gather.csv, profiles/%s_pmc_fetch_size.csv, profiles/%s_pmc_write_size.csv" % (TAG, TAG, TAG), **BUILD,
    }, open(os.path.join(DST, "viterbi_counters.json"), "w"), indent=1)
    print(open(os.path.join(DST, "viterbi_counters.json")).read())
if os.path.exists(os.path.join(SRC, "valu_rate.txt")):
    shutil.copy(os.path.join(SRC, "valu_rate.txt"), os.path.join(DST, TAG + "_ubench_valu_rate.txt"))
    cyc = {}
    for line in open(os.path.join(SRC, "valu_rate.txt")):
        if "waves/SIMD 5:" in line:
            name = line.split("waves/SIMD")[0].strip(); cyc[name] = float(line.split("->")[1].split()[0])
    if cyc:
        packed = [cyc[k] for k in ("v_pk_add_u16", "v_pk_min_u16", "v_pk_sub_i16", "v_perm_b32 (vgpr sel)", "v_and_or_b32", "v_pk_add_u16 op_sel") if k in cyc]
        cp = sum(packed) / len(packed); cl = cyc.get("v_add_u32", 2.4)
        # instruction mix of one trellis step (viterbi_acs.h; five of six layouts add with plain 32-bit additions): packed 16-bit min /
        # sub and the decision extraction (v_perm_b32 / v_and_or_b32) 96 per step, plus 64 packed additions in the sixth layout -> 107
        # on average; plain: 64 * 5 / 6 = 53 additions + about 35 for branch metrics, addresses and the loop
        n_packed, n_plain = (5 * 96 + 160) / 6.0, 64 * 5 / 6.0 + 35
        json.dump({"cycles_packed": cp, "cycles_plain": cl, "cycles_per_instruction_kernel_mix": (n_packed * cp + n_plain * cl) / (n_packed + n_plain), "at_waves_per_simd": 5, "per_instruction": cyc,
                   "source": "profiles/%s_ubench_valu_rate.txt (tools/ubench/valu_rate.hip on the GPU box, 2.4 GHz assumed)" % TAG, **BUILD},
                  open(os.path.join(DST, "valu_rate.json"), "w"), indent=1)


# ---- profiles/<tag>_summary.md: the figures DESIGN.md quotes, derived from the files above and from the bench line of the same session
# (gpurun_out/prof/bench.json when make_profiles.sh left one), so that no prose number is typed by hand
def _kernel_stats():
    out = {}
    for r in csv.DictReader(open(os.path.join(DST, TAG + "_bench_kernel_stats.csv"))):
        name = r.get("Name") or r.get("Kernel_Name") or ""
        if "dabphy" not in name:
            continue
        k = name.split("(")[0].replace("dabphy::", "").replace("void ", "")
        out[k] = dict(calls=int(r["Calls"]), avg_ms=float(r["AverageNs"]) / 1e6, min_ms=float(r["MinNs"]) / 1e6, max_ms=float(r["MaxNs"]) / 1e6, pct=float(r["Percentage"]))
    return out


try:
    ks = _kernel_stats()
    dj = json.load(open(os.path.join(DST, "demod_hbm_traffic.json")))
    vj = json.load(open(os.path.join(DST, "viterbi_counters.json"))) if os.path.exists(os.path.join(DST, "viterbi_counters.json")) else {}
    lines = ["# %s profile summary (generated by tools/collect_profiles.py -- do not edit)" % TAG, "",
             "Build: src_sha256 `%s...`, lib_sha256 `%s...`; batch %d ensembles x %d frames; command: `tools/make_profiles.sh` (rocprofv3 --kernel-trace --stats of the default" % (BUILD["src_sha256"][:16], BUILD["lib_sha256"][:16], B, F),
             "`bench.py` step counts; counters in their own --pmc passes).", "", "## Kernel statistics (`%s_bench_kernel_stats.csv`)" % TAG, "",
             "| kernel | launches | average ms | minimum ms | maximum ms | % of kernel time |", "|---|---|---|---|---|---|"]
    for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["pct"]):
        lines.append("| `%s` | %d | %.4f | %.4f | %.4f | %.2f |" % (k, v["calls"], v["avg_ms"], v["min_ms"], v["max_ms"], v["pct"]))
    kd = [k for k in ks if k.startswith("k_demod")]
    if kd:
        v = ks[kd[0]]; alg = dj["algorithmic_bytes_per_launch"]
        lines += ["", "## FFT stage (`k_demod`) against the HBM roofline", "",
                  "* algorithmic bytes per launch: %d (IQ useful parts read once + int8 soft bits written once)" % alg,
                  "* average launch %.4f ms -> %.0f GB/s = **%.3f of 8 TB/s**; minimum launch %.4f ms -> %.3f" % (v["avg_ms"], alg / v["avg_ms"] / 1e6, alg / v["avg_ms"] / 1e6 / 8000.0, v["min_ms"], alg / v["min_ms"] / 1e6 / 8000.0),
                  "* HBM traffic from the counters: FETCH_SIZE %.0f KB x 2 (gfx950 correction) + WRITE_SIZE %.0f KB = %d bytes = **%.3f x algorithmic**" % (dj["fetch_size_kb_raw"], dj["write_size_kb_raw"], dj["hbm_bytes_per_launch"], dj["traffic_over_algorithmic"])]
    kv = [k for k in ks if k.startswith("k_viterbi_fused")]
    if kv and vj:
        v = ks[kv[0]]
        n_simd = 1024
        lines += ["", "## Viterbi stage (`k_viterbi_fused`, all classes + FIC in one launch)", "",
                  "* average launch %.4f ms (minimum %.4f)" % (v["avg_ms"], v["min_ms"]),
                  "* SQ_INSTS_VALU %.4g per launch -> %.4g wave-instructions/s = **%.3f** of one plain instruction per SIMD every two cycles (%d SIMDs x 2.4 GHz / 2)" % (vj["valu_insts_per_launch"], vj["valu_insts_per_launch"] / (v["avg_ms"] * 1e-3), vj["valu_insts_per_launch"] / (v["avg_ms"] * 1e-3) / (n_simd * 2.4e9 / 2), n_simd),
                  "* HBM: FETCH_SIZE %.0f KB raw, WRITE_SIZE %.0f KB -> %.2f GB with the fetch doubled, %.2f GB with narrow requests tallied in full, against %.2f GB algorithmic (%.1f x / %.1f x); decision array written + read: %.2f GB" % (
                      vj["fetch_size_kb_raw"], vj["write_size_kb_raw"], vj["hbm_bytes_per_launch"] / 1e9, vj["hbm_bytes_if_narrow_requests_are_tallied_in_full"] / 1e9, vj["algorithmic_bytes_per_launch"] / 1e9,
                      vj["hbm_bytes_per_launch"] / vj["algorithmic_bytes_per_launch"], vj["hbm_bytes_if_narrow_requests_are_tallied_in_full"] / vj["algorithmic_bytes_per_launch"], vj["decision_bytes_written_plus_read"] / 1e9),
                  "* LDS bank efficiency 1 - SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = %.3f" % (1.0 - vj["lds_bank_conflict_cycles"] / vj["lds_idx_active_cycles"] if vj.get("lds_idx_active_cycles") else float("nan"))]
    bj = os.path.join(SRC, "bench.json")
    if os.path.exists(bj):
        try:
            j = json.loads([l for l in open(bj).read().splitlines() if l.startswith("{")][-1])
            shutil.copy(bj, os.path.join(DST, TAG + "_bench_default.json"))
            lines += ["", "## The bench line of the same session (`%s_bench_default.json`)" % TAG, "",
                      "* value **%.0f x real-time**, %.3f ms per step; stages (HIP events, ms): %s" % (j["value"], j["ms_per_step"], ", ".join("%s %.3f" % kv for kv in j["stages_ms"].items())),
                      "* roofline.frac %.4f (kernel_ms %.4f); measured float4 copy %.0f GB/s -> %.3f of achievable" % (j["roofline"]["frac"], j["roofline"]["kernel_ms"], j["roofline"].get("measured_copy_GBps", float("nan")), j["roofline"].get("frac_of_achievable", float("nan")))]
            cb = j.get("cpu_baseline")
            if cb:
                lines.append("* cpu_baseline: %.1f x real-time (%s, %d cores); oracle receivers on all cores %.1f x; -O3 build %s" % (cb["value"], cb["kind"], cb["cores"], (cb.get("oracle_port") or {}).get("value", float("nan")), ("%.1f x" % cb["o3"]["value"]) if cb.get("o3") and "value" in cb["o3"] else "-"))
            ex = j.get("extras", {})
            for name in ("hetero", "mixed_layouts"):
                e = ex.get(name)
                if e and "value" in e:
                    lines.append("* extras.%s: %.0f x, %.3f ms per step, decoder %.3f ms, superframe filter %.3f ms, parity %s" % (name, e["value"], e["ms_per_step"], e["msc_viterbi_ms"], e["stages_ms"].get("rs", float("nan")), e.get("parity")))
            for name in ("drift", "low_snr"):
                e = ex.get(name)
                if e and "value" in e:
                    w = e.get("wide_sync_stats", {})
                    lines.append("* extras.%s: %.0f x (%.3f of the headline), %.3f ms per step (demod %.3f, decoder %.3f, filter %.3f, synchroniser chain %.3f from its gate); frames accepted from the wide pass / find chain %s of %s, passes that needed the serial chain %s of %s, batches decoded twice %s; Reed-Solomon corrected %s symbols; parity %s%s" % (
                        name, e["value"], e["value"] / j["value"], e["ms_per_step"], e["stages_ms"].get("demod", float("nan")), e["stages_ms"].get("msc_viterbi", float("nan")), e["stages_ms"].get("rs", float("nan")), e["stages_ms"].get("sync", float("nan")),
                        w.get("frames_accepted_from_the_wide_pass"), w.get("frames"), w.get("passes_that_needed_the_serial_chain"), w.get("passes"), e.get("replayed_batches"), (e.get("superframes") or {}).get("rs_corrected_symbols"), e.get("parity"),
                        ("; synchroniser always behind the decoder (sync_early = 1): %.3f ms per step" % e["synchroniser_always_behind_the_decoder"]["ms_per_step"]) if "ms_per_step" in (e.get("synchroniser_always_behind_the_decoder") or {}) else ""))
            md = j.get("msc_drain")
            if md and "value" in md:
                lines.append("* msc_drain: every step ALL %d services' logical frames on the host (%d bytes, %d frames) through the bulk drain overlapped with the next step: %.3f ms per step = %.0f x (%.3f of the headline), %.1f GB/s into page-locked memory; the drain alone %.3f ms (%.1f GB/s)" % (
                    md["services"], md["bytes_per_step"], md["logical_frames_per_step"], md["ms_per_step_all_services_on_the_host"], md["value"], md["vs_headline"], md["host_GBps"], md["drain_alone_ms"], md["drain_alone_GBps"]))
            fc = j.get("facade")
            if fc and "level2" in fc:
                for k, v in fc["level2"].items():
                    if "cpu_ms_per_frame" in v:
                        a = v.get("all_18_services", {})
                        lines.append("* level 2 `%s`: %.2f ms wall / %.2f ms CPU per frame with two services%s" % (k, v["ms_per_frame"], v["cpu_ms_per_frame"],
                                     ("; all 18 services: %.2f wall / %.2f CPU%s" % (a["ms_per_frame"], a["cpu_ms_per_frame"], (", %.1f code words per device call" % a["code_words_per_device_call"]) if "code_words_per_device_call" in a else "")) if "ms_per_frame" in a else ""))
            if fc and "ms_per_frame" in fc:
                lines.append("* facade: %.3f ms per 96 ms frame; level 2 builds (ms per frame): %s" % (fc["ms_per_frame"], ", ".join("%s %.2f" % (k, v["ms_per_frame"]) for k, v in fc.get("level2", {}).items() if "ms_per_frame" in v)))
        except Exception as ex_:
            lines.append("(bench line not summarised: %s)" % ex_)
    # the -m gpu suite of the same device session (GPUTEST_LOG = its pytest output), e.g. gpurun_out/r5g/gputest.log; GPUTEST_NOTE = what to
    # say about it when the log is not a full suite run of this very session
    gl = os.environ.get("GPUTEST_LOG")
    if gl and os.path.exists(gl):
        res = [l.strip() for l in open(gl).read().splitlines() if " passed" in l or " failed" in l or l.startswith("pytest rc")]
        note = os.environ.get("GPUTEST_NOTE")
        lines += ["", "## The `-m gpu` %s (`%s`)" % ("suite of the same session" if not note else "tests: " + note, gl), "", "* " + "; ".join(res[-2:])]
    for k in range(2, 4):                               # further logs (GPUTEST_LOG2 / GPUTEST_NOTE2, ...)
        gl = os.environ.get("GPUTEST_LOG%d" % k)
        if gl and os.path.exists(gl):
            res = [l.strip() for l in open(gl).read().splitlines() if " passed" in l or " failed" in l or l.startswith("pytest rc")]
            lines += ["", "## The `-m gpu` tests: %s (`%s`)" % (os.environ.get("GPUTEST_NOTE%d" % k, ""), gl), "", "* " + "; ".join(res[-2:])]
    open(os.path.join(DST, TAG + "_summary.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
except Exception as ex_:
    print("summary not written:", ex_)
