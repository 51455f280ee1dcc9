#!/usr/bin/env python
"""bench.py -- DAB Mode-I PHY hot path on MI355X: whole-job throughput, FFT-stage HBM roofline, Viterbi issue roofline, CPU baseline.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.  Started without a
torch.distributed environment and N > 1 it launches the N ranks itself (torch.distributed.run, 127.0.0.1); started by the driver's
own torch.distributed.run it is one of the ranks.  Every rank decodes its own block of ensembles (the path shards by ensemble, no
data-path collective); the decoded FIBs + CRC flags are gathered to rank 0 over RCCL per step, device buffer to device buffer.

A "step" is one pass of the whole hot path (time/frequency sync, NCO, 2048-pt FFT, DQPSK demap, de-interleavers, depuncture,
Viterbi, energy dispersal, FIB CRC, superframe filter with Reed-Solomon) over one batch of B ensembles x F transmission frames of
synthetic 2.048 Msps complex-float IQ that is already resident in HBM (BASELINE.json config "1xMI355X: batch of 256 synthetic Mode-I
ensembles"; welle_io_amd/workload.py builds signal and handle, tests/test_gpu_bench_config.py checks exactly this configuration
against the oracle).  value = ensembles decoded in real time (x real-time) summed over all GPUs = N*B*F*0.096 s / step time.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package, PKG_DIR  # noqa: E402

FRAME_S = 0.096
T_F = 196608
ALG_BYTES_DEMOD_PER_FRAME = 76 * 2048 * 8 + 75 * 3072      # IQ useful parts read once + int8 soft bits written once
ALG_BYTES_FFT_CLASSIC_PER_FRAME = 76 * 2 * 2048 * 8          # SURVEY 8(d) c2c accounting (read + write of each FFT)
HBM_PEAK_GBPS = 8000.0                                       # MI355X_MICROARCH.md: 8 TB/s spec
CLOCK_HZ = 2.4e9                                             # MI355X_MICROARCH.md: peak engine clock
PARITY_SUBCH = (0, 7, 17)                                    # sub-channels whose MSC bytes the parity leg compares


def _run_receivers(recs, n_proc, n_loops, mode, env, out_dir, cwd=None, layout=None):
    """n_proc concurrent receiver processes, receiver i over recording recs[i % len(recs)]; returns their result records
    (layout: JSON file with the multiplex when it is not the canonical one)"""
    args = [sys.executable, os.path.join(ROOT, "tests", "cpu_baseline_worker.py")]
    procs = [subprocess.Popen(args + [recs[i % len(recs)], str(n_loops), mode, os.path.join(out_dir, "%s_%d.npz" % (mode, i)) if i < len(recs) else "-"] + ([layout] if layout else []),
                              stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=open(os.path.join(out_dir, "%s_err.txt" % mode), "w") if i == 0 else subprocess.DEVNULL, text=True, env=env, cwd=cwd) for i in range(n_proc)]
    for p in procs:
        if p.stdout.readline().strip() != "READY":
            err = open(os.path.join(out_dir, "%s_err.txt" % mode)).read()[-400:]
            for q in procs:
                q.kill()
            raise RuntimeError("CPU baseline worker did not start " + err)
    for p in procs:
        p.stdin.write("go\n"); p.stdin.flush()            # all receivers start together
    res = [json.loads(p.stdout.readline()) for p in procs]
    for p in procs:
        p.wait()
    return res


def _compare_with_receivers(td, mode, ens_list, gpu_logs, subch_idx=PARITY_SUBCH):
    """FIB bytes + CRC flags and the MSC bytes of the sub-channels subch_idx of every ensemble in ens_list: GPU log vs the files receiver i
    of `mode` left in td; raises AssertionError on the first difference; returns (frames compared, MSC bytes compared of the last sub-channel)"""
    n = m = 0
    for i, e in enumerate(ens_list):
        z = np.load(os.path.join(td, "%s_%d.npz" % (mode, i)))
        g = gpu_logs[e]
        n = min(len(g["fib"]), len(z["fib"]) // 12)
        assert n >= len(g["fib"]) - 1 and n > 0, (mode, e, n, len(g["fib"]))
        zf = z["fib"][:12 * n].reshape(n, 12, 33)
        ok = np.array_equal(np.array(g["ok"][:n]), zf[:, :, 0]) and np.array_equal(np.array(g["fib"][:n]), zf[:, :, 1:])
        if not ok:
            raise AssertionError("parity: FIBs of ensemble %d differ from the %s receiver's" % (e, mode))
        for k, i_sub in enumerate(subch_idx):
            got = b"".join(g["msc"][k]); want = z["msc%d" % i_sub].tobytes()
            m = min(len(got), len(want))
            if m == 0 or got[:m] != want[:m]:
                raise AssertionError("parity: MSC bytes of ensemble %d sub-channel %d differ from the %s receiver's" % (e, i_sub, mode))
    return n, m


def _host_has(*flags):
    try:
        have = set(open("/proc/cpuinfo").read().split("flags", 1)[1].split("\n", 1)[0].split())
    except Exception:
        return False
    return all(f in have for f in flags)


def _stage_ms_from_profile(path, frames):
    """profiling_points.csv of the reference's -DWITH_PROFILING build (thread, mark, CLOCK_THREAD_CPUTIME_ID stamp per PROFILE() call,
    various/profiling.cpp:183-188): the CPU time between a mark and the next one on the same thread is booked on the FIRST of the two
    (as the reference's own profiling.dot does, :150-160) -> CPU milliseconds per transmission frame and mark"""
    import csv
    last = {}; acc = {}
    with open(path) as fh:
        for r in csv.DictReader(fh):
            t = int(r["time_sec"]) + int(r["time_ns"]) * 1e-9
            th = r["thread_id"]
            if th in last:
                acc[last[th][0]] = acc.get(last[th][0], 0.0) + (t - last[th][1])
            last[th] = (r["mark"], t)
    return {k: v * 1e3 / max(1, frames) for k, v in sorted(acc.items())}


def cpu_baseline(rows, n_loops, gpu_logs):
    """The CPU side of the same workload on this host, a bounded sample: the very rows of the GPU batch named in `rows` (ensemble index
    -> looping recording incl. its noise and carrier offset), one receiver per ensemble as the reference runs them, all 18 sub-channels:
      * kind "reference": the REAL reference backend (oracle/_ref, built from /root/reference where that exists) -- RadioReceiver
        with its own 2-3 threads and SSE Viterbi; it also runs the layers above the PHY (FIG parsing, superframe filter, AAC).
        cores/2 receivers run concurrently so that every core is busy.
      * the oracle (C restatement of the PHY only, single-threaded), one receiver per core, as a second figure (or as the
        baseline, kind "port", where oracle/_ref is absent).
    The receivers' outputs for the first frames are also the PARITY CHECK of this very run: FIB bytes + CRC flags and the MSC bytes of
    three sub-channels of those ensembles, GPU (logged during warm-up) vs CPU, byte for byte."""
    import tempfile
    cores = max(1, min(32, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    ens = sorted(rows)
    with tempfile.TemporaryDirectory() as td:
        recs = []
        for e in ens:
            path = os.path.join(td, "rec%d.npy" % e); np.save(path, rows[e]); recs.append(path)
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        res = _run_receivers(recs, cores, n_loops, "port", env, td)
        ref = None; ref_error = None; o3 = None; stage_ms = None
        have_ref = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libwelle_ref.so"))
        # the reference leg decodes the same rows as the oracle leg (the synthetic access units open with ID_END, so its FAAD2 adapter
        # rejects them instead of half-decoding random bytes and throwing: synth.make_superframe)
        if have_ref:
            try:
                ref = _run_receivers(recs, max(len(recs), cores // 2), max(1, n_loops // 2), "reference", env, td)
            except Exception as ex:                        # reported in the line (kind falls back to "port")
                ref = None; ref_error = "%s: %s" % (type(ex).__name__, ex)
        # ---- parity of this run: GPU log vs CPU receivers on the same rows
        parity = {"ensembles": ens, "sub_channels": list(PARITY_SUBCH), "against": []}
        for mode in (["reference"] if ref else []) + ["port"]:
            n, m = _compare_with_receivers(td, mode, ens, gpu_logs)
            parity["frames"] = n; parity["msc_bytes_per_sub_channel"] = m
            parity["against"].append(("reference backend (oracle/_ref), ensembles %s" if mode == "reference" else "oracle (C restatement), ensembles %s") % ens)
        parity["fib_equal"] = True; parity["msc_equal"] = True
        # ---- SURVEY 8(d)'s extras: the -O3 courtesy build of the reference, and its own PROFILE() marks (one receiver, PHY stages)
        if ref and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libwelle_ref_o3.so")) and _host_has("avx2", "fma", "bmi2"):
            try:
                r3 = _run_receivers(recs, max(len(recs), cores // 2), max(1, n_loops // 2), "reference_o3", env, td)
                o3 = dict(value=sum(r["frames"] for r in r3) * FRAME_S / max(r["seconds"] for r in r3), per_receiver=r3[0]["frames"] * FRAME_S / r3[0]["seconds"],
                          flags="-O3 -march=x86-64-v3 (built on the build machine: not -march=native of this host)", receivers=len(r3))
            except Exception as ex:
                o3 = {"error": "%s: %s" % (type(ex).__name__, ex)}
        if ref and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libwelle_ref_prof.so")):
            try:
                pd = os.path.join(td, "prof"); os.makedirs(pd)
                rp = _run_receivers(recs[:1], 1, max(1, n_loops // 2), "reference_prof", env, td, cwd=pd)
                stage_ms = _stage_ms_from_profile(os.path.join(pd, "profiling_points.csv"), rp[0]["frames"])
            except Exception as ex:
                stage_ms = {"error": "%s: %s" % (type(ex).__name__, ex)}
    port_value = sum(r["frames"] for r in res) * FRAME_S / max(r["seconds"] for r in res)
    port = dict(value=port_value, per_core=res[0]["frames"] * FRAME_S / res[0]["seconds"], receivers=cores,
                fib_ok=sum(r["fib_ok"] for r in res), fibs=sum(r["fibs"] for r in res))
    if ref:
        slowest = max(r["seconds"] for r in ref)
        base = dict(value=sum(r["frames"] for r in ref) * FRAME_S / slowest, unit="x real-time (ensembles decoded concurrently by all cores)", cores=cores,
                    kind="reference", per_receiver=ref[0]["frames"] * FRAME_S / ref[0]["seconds"],
                    sample="%d concurrent RadioReceivers of the reference backend (each 2-3 threads; PHY + FIG parsing + superframe filter + AAC; lock-step input so that no frame is dropped) x %d frames "
                           "(%.1f s of IQ each) of ensembles %s of this batch, 18 sub-channels, slowest receiver %.1f s"
                           % (len(ref), ref[0]["frames"], ref[0]["frames"] * FRAME_S, ens, slowest),
                    fib_ok=sum(r["fib_ok"] for r in ref), fibs=sum(r["fibs"] for r in ref), oracle_port=port, o3=o3,
                    stage_ms=stage_ms, stage_ms_note="CPU ms per transmission frame between the reference's own PROFILE() marks (various/profiling.h:39-62; -DWITH_PROFILING build of the same sources, ONE receiver alone on the host, thread CPU time): OFDMProcessor thread NotSynced..DecodeTII, OfdmDecoder thread ProcessPRS..SymbolProcessed, one DabAudio thread per sub-channel DAGetMSCData..DADone (18 of them: their marks are summed)")
    else:
        base = dict(value=port_value, unit="x real-time (ensembles decoded concurrently by all cores)", cores=cores, kind="port",
                    per_core=port["per_core"],
                    sample="%d receivers x %d frames (%.1f s of IQ each) of ensembles %s of this batch, 18 sub-channels, oracle C restatement, slowest receiver %.1f s"
                           % (cores, res[0]["frames"], res[0]["frames"] * FRAME_S, ens, max(r["seconds"] for r in res)),
                    fib_ok=port["fib_ok"], fibs=port["fibs"], reference_error=ref_error)
    return base, parity


def hetero_leg(capi, workload, torch, lib_path, B, F, steps, local, sched):
    """The same hot path over a multiplex as they are on air (workload.HETERO_LAYOUT: 15 sub-channels, 6 protection classes incl. EEP-B
    and UEP, code words of 192 .. 3072 bits), B x F, timed after the headline; never `value`.  Parity of this very run: FIBs and the MSC
    bytes of EVERY sub-channel of the first, a middle and the last ensemble against the oracle on the same rows."""
    import tempfile
    lib = capi.load_library(lib_path)
    subchs = workload.hetero_subchannels(lib)
    rec_frames = workload.rec_frames_for(F)
    base = workload.make_base_streams(2, rec_frames, seed0=50, subchs=subchs)
    iq, cfo_hz, base_np, txs = workload.make_batch(B, rank=7, device="cuda", base=base, rec_frames=rec_frames)
    dev = workload.open_receiver(capi, lib_path, iq, F, subchs, device=local, pipeline_sync=sched)
    out = {"workload": "%d ensembles x %d frames, %d sub-channels in %d protection classes: %s" % (
        B, F, len(subchs), len({(s.bitrate, s.profile_b, s.level, s.uep is not None) for s in subchs}),
        ", ".join("%dk %s" % (s.bitrate, ("UEP-%d" % s.level) if s.uep is not None else "EEP-%d%s" % (s.level, "B" if s.profile_b else "A")) for s in subchs))}
    try:
        check = sorted({0, B // 2, B - 1})
        idx = list(range(len(subchs)))
        logs = {e: dict(fib=[], ok=[], msc=[[] for _ in idx]) for e in check}
        for W in range(3):
            dev.process(F); sf = dev.superframes_stats()
            if W * F < 64:
                info = dev.frame_info(); fib, ok = dev.fibs(); mscs = [dev.msc(i) for i in idx]
                for e in check:
                    valid = [f for f in range(F) if info[e, f]["valid"] == 1]
                    for f in valid:
                        logs[e]["fib"].append(np.array(fib[e, f])); logs[e]["ok"].append(np.array(ok[e, f]))
                    for k in idx:
                        m, fv = mscs[k]
                        logs[e]["msc"][k].append(m[e, fv[e]:4 * len(valid)].tobytes())
        torch.cuda.synchronize(); t0 = time.perf_counter(); acc = {}
        for _ in range(steps):
            dev.process(F); sf = dev.superframes_stats(); fib, ok = dev.fibs_host()
            for k, v in dev.stage_times().items():
                acc[k] = acc.get(k, 0.0) + v
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        assert np.asarray(ok).all(), "FIB CRC failures in the heterogeneous signal"
        assert (sf[:, 2] == 0).all() and (sf[:, 3] == 0).all() and (sf[:, 0] >= len(subchs) * (4 * F // 5)).all(), "superframe filter: %s" % sf[:4]
        n_cw_steps = B * 4 * F * sum(24 * s.bitrate + 6 for s in subchs)
        out.update(value=B * F * FRAME_S / dt, unit="x real-time", ms_per_step=dt * 1e3, stages_ms={k: v / steps for k, v in acc.items()},
                   msc_viterbi_ms=acc.get("msc_viterbi", 0.0) / steps, demod_ms=acc.get("demod", 0.0) / steps,
                   codeword_steps_per_s=n_cw_steps / (acc.get("msc_viterbi", 0.0) / steps * 1e-3) if acc.get("msc_viterbi") else None,
                   launches="one fused launch for all classes and the FIC (dabphy_fused.hip: work list, longest code words first)")
        with tempfile.TemporaryDirectory() as td:
            recs = []
            for e in check:
                path = os.path.join(td, "rec%d.npy" % e); np.save(path, iq[e].cpu().numpy()); recs.append(path)
            layout = os.path.join(td, "layout.json"); json.dump(workload.subchannels_to_json(subchs), open(layout, "w"))
            env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
            _run_receivers(recs, len(recs), max(2, -(-80 // rec_frames)), "port", env, td, layout=layout)
            n, m = _compare_with_receivers(td, "port", check, logs, subch_idx=idx)
        out.update(parity=True, parity_detail="FIBs + CRC flags of %d frames and the MSC bytes of all %d sub-channels of ensembles %s equal the oracle's (C restatement, same rows)" % (n, len(subchs), check))
    except AssertionError as ex:
        out.update(parity=False, parity_error=str(ex))
    finally:
        dev.close()
    return out


def mixed_layouts_leg(capi, workload, torch, lib_path, B, F, steps, local, sched):
    """A batch of INDEPENDENT ensembles, what a rack of receivers is (every receiver of the reference selects its own services,
    msc-handler.cpp:61-127): ensemble b receives multiplex b % 5 of workload.mixed_layouts (canonical, heterogeneous, two random ones,
    canonical) and selects ITS sub-channels (all / all / all / every other service / none) through dabphy_set_subchannels_ensemble.
    B x F, timed after the headline; never `value`.  Parity of this very run: FIBs and the MSC bytes of EVERY selected sub-channel of one
    ensemble per layout and of the last ensemble against the oracle on the same rows."""
    import tempfile
    lib = capi.load_library(lib_path)
    tx_lists, sel_lists = workload.mixed_layouts(lib)
    nd = len(tx_lists)
    rec_frames = workload.rec_frames_for(F)
    base = workload.make_base_streams(nd, rec_frames, seed0=70, subchs=tx_lists)
    iq, cfo_hz, base_np, txs = workload.make_batch(B, rank=9, device="cuda", base=base, rec_frames=rec_frames)
    sel = [sel_lists[b % nd] for b in range(B)]
    dev = workload.open_receiver(capi, lib_path, iq, F, sel, device=local, pipeline_sync=sched)
    classes = {(s.bitrate, s.profile_b, s.level, s.uep is not None) for l in sel for s in l}
    out = {"workload": "%d ensembles x %d frames, ensemble b on multiplex b %% %d with its own selection: %s sub-channels selected (of %s transmitted), %d protection classes over the batch"
                       % (B, F, nd, "/".join(str(len(l)) for l in sel_lists), "/".join(str(len(l)) for l in tx_lists), len(classes)),
           "selected_subchannels": sum(len(l) for l in sel)}
    try:
        check = sorted(set(range(min(nd, B))) | {B - 1})
        logs = {e: dict(fib=[], ok=[], msc=[[] for _ in sel[e]]) for e in check}
        for W in range(3):
            dev.process(F); sf = dev.superframes_stats()
            if W * F < 64:
                info = dev.frame_info(); fib, ok = dev.fibs()
                for e in check:
                    valid = [f for f in range(F) if info[e, f]["valid"] == 1]
                    for f in valid:
                        logs[e]["fib"].append(np.array(fib[e, f])); logs[e]["ok"].append(np.array(ok[e, f]))
                    for k in range(len(sel[e])):
                        m, fv, nr = dev.msc_ensemble(e, k)
                        logs[e]["msc"][k].append(m[fv:nr].tobytes())
        torch.cuda.synchronize(); t0 = time.perf_counter(); acc = {}
        for _ in range(steps):
            dev.process(F); sf = dev.superframes_stats(); fib, ok = dev.fibs_host()
            for k, v in dev.stage_times().items():
                acc[k] = acc.get(k, 0.0) + v
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        assert np.asarray(ok).all(), "FIB CRC failures in the mixed-layout signal"
        want_sf = np.array([len(l) * (4 * F // 5) for l in sel])
        assert (sf[:, 2] == 0).all() and (sf[:, 3] == 0).all() and (sf[:, 0] >= want_sf).all() and (sf[want_sf == 0, 0] == 0).all(), "superframe filter: %s" % sf[:5]
        n_cw_steps = 4 * F * sum(24 * s.bitrate + 6 for l in sel for s in l)
        out.update(value=B * F * FRAME_S / dt, unit="x real-time", ms_per_step=dt * 1e3, stages_ms={k: v / steps for k, v in acc.items()},
                   msc_viterbi_ms=acc.get("msc_viterbi", 0.0) / steps, demod_ms=acc.get("demod", 0.0) / steps,
                   codeword_steps_per_s=n_cw_steps / (acc.get("msc_viterbi", 0.0) / steps * 1e-3) if acc.get("msc_viterbi") else None,
                   launches="one fused launch for all classes of all ensembles and the FIC (dabphy_fused.hip: classes = lists of (ensemble, sub-channel) pairs)")
        n = 0
        for e in check:
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "rec.npy"); np.save(path, iq[e].cpu().numpy())
                layout = os.path.join(td, "layout.json"); json.dump(workload.subchannels_to_json(sel[e]), open(layout, "w"))
                env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
                _run_receivers([path], 1, max(2, -(-80 // rec_frames)), "port", env, td, layout=layout)
                if sel[e]:
                    n, m = _compare_with_receivers(td, "port", [e], logs, subch_idx=list(range(len(sel[e]))))
                else:
                    z = np.load(os.path.join(td, "port_0.npz")); g = logs[e]
                    k = min(len(g["fib"]), len(z["fib"]) // 12); zf = z["fib"][:12 * k].reshape(k, 12, 33)
                    assert k > 0 and np.array_equal(np.array(g["ok"][:k]), zf[:, :, 0]) and np.array_equal(np.array(g["fib"][:k]), zf[:, :, 1:]), "parity: FIBs of ensemble %d differ from the oracle's" % e
        out.update(parity=True, parity_detail="FIBs + CRC flags of %d frames and the MSC bytes of every selected sub-channel of ensembles %s (one per layout, and the last) equal the oracle's (C restatement, same rows, same selection)" % (n, check))
    except AssertionError as ex:
        out.update(parity=False, parity_error=str(ex))
    finally:
        dev.close()
    return out


def msc_drain_leg(dev, torch, B, F, steps, n_services):
    """What the headline leaves in HBM: the decoded logical frames of EVERY service (B x 18 sub-channels x 4F frames x 192 bytes).  Here they
    all reach the host, every step, through the bulk drain (dabphy_msc_drain_begin: one device-to-host copy per protection class into
    page-locked memory + an index table; msc-handler.cpp:129-158 / dab-audio.cpp:151-160 for a whole batch) OVERLAPPED with the next step:
    the drain of batch k is in flight while batch k + 1 is synchronised and demodulated, its decoders wait for it on the device.  Timed on
    the headline's handle right after the timed region; never `value`."""
    nb, nd = dev.msc_batch_size()
    assert nd == n_services, (nd, n_services)
    pinned = dev.host_alloc((nb,), np.uint8)
    desc = None                                                   # (the binding allocates the index table)
    out = {}
    try:
        # the drain alone: the device otherwise idle
        t = []
        for _ in range(3):
            t0 = time.perf_counter(); dev.msc_batch(pinned, desc); t.append(time.perf_counter() - t0)
        out["drain_alone_ms"] = min(t) * 1e3; out["drain_alone_GBps"] = nb / min(t) / 1e9
        # overlapped with the steps
        dev.process(F); dev.superframes_stats()
        torch.cuda.synchronize(); t0 = time.perf_counter(); acc = {}; t_begin = t_wait = 0.0
        for _ in range(steps):
            ta = time.perf_counter()
            _, dsc = dev.msc_drain_begin(pinned, desc)            # batch k leaves ...
            tb = time.perf_counter()
            dev.process(F); dev.superframes_stats(); dev.fibs_host()      # ... while batch k + 1 is decoded
            tc = time.perf_counter()
            dev.msc_drain_wait()                                  # all services of batch k are on the host
            t_begin += tb - ta; t_wait += time.perf_counter() - tc
            for k, v in dev.stage_times().items():
                acc[k] = acc.get(k, 0.0) + v
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        out.update(stages_ms={k: v / steps for k, v in acc.items()}, host_ms_in_drain_begin=t_begin / steps * 1e3, host_ms_in_drain_wait=t_wait / steps * 1e3)
        rows = int((dsc["n_rows"] - dsc["first_valid"]).sum())
        out.update(ms_per_step_all_services_on_the_host=dt * 1e3, value=B * F * FRAME_S / dt, unit="x real-time", services=int(nd), bytes_per_step=int(nb),
                   logical_frames_per_step=rows, host_GBps=nb / dt / 1e9,
                   what="every step: dabphy_msc_drain_begin (one copy per protection class into page-locked memory) -> the next dabphy_process -> dabphy_msc_drain_wait; the headline's handle, after its timed region")
    finally:
        dev.msc_drain_wait(); dev.host_free(pinned)
    return out


def channel_leg(capi, workload, torch, lib_path, B, F, steps, local, sched, kind, device="cuda"):
    """The headline's batch geometry on signals as a receiver MEETS them; timed after the headline, never `value`.
      kind "drift":   every ensemble has its own sampling-clock offset, +-1 ... +-50 ppm (log-uniform), sigma 0.02, a NON-looping stream:
                      the PRS window index moves from frame to frame (ofdm-processor.cpp:337-350,432-463; phasereference.cpp:73-256),
                      so the wide synchroniser pass's prediction (window index T_g, correctors unmoved) fails and the frame-by-frame
                      chain takes over;
      kind "low_snr": every ensemble has its own SNR, 6 ... 10 dB (looping recordings as in the headline): Reed-Solomon corrects and
                      sometimes gives up, FIBs fail their CRC, fine correctors move, batches may be decoded twice (exact batch mode).
    Reports x real-time, the stage times, how the synchroniser's wide pass fared and the batches decoded twice.  Parity of this very run:
    FIBs + CRC flags and the MSC bytes of three sub-channels of four ensembles against the oracle on the same rows."""
    import tempfile
    sync_ = torch.cuda.synchronize if device == "cuda" else (lambda: None)      # (device "cpu": tests/test_workload.py runs the leg on the kernels' CPU execution model)
    rec_frames = workload.rec_frames_for(F)
    base = workload.make_base_streams(min(4, B), rec_frames, seed0=300)
    subchs = base[1][0].subchs
    n_warm = 3
    if kind == "drift":
        steps = max(2, min(steps, 5))
        n_frames = (n_warm + steps + 2) * F + 4                # the synchroniser runs a batch ahead and wants a whole frame beyond it
        iq, par, base_np, txs = workload.make_channel_batch(B, n_frames, rank=11, ppm=(1.0, 50.0), device=device, base=base, rec_frames=rec_frames)
        loop = False
        what = "sampling-clock offset log-uniform in +-[1, 50] ppm per ensemble (non-looping stream of %d frames, band-limited resampling), sigma 0.02" % n_frames
    else:
        iq, par, base_np, txs = workload.make_channel_batch(B, rec_frames, rank=12, snr_db=(6.0, 10.0), device=device, base=base, rec_frames=rec_frames)
        loop = True
        what = "SNR uniform in [6, 10] dB per ensemble (looping recordings, noise part of the loop)"
    dev = workload.open_receiver(capi, lib_path, iq, F, subchs, device=local, pipeline_sync=sched, loop=loop)
    out = {"workload": "%d ensembles x %d frames, 18 x 64 kbit/s DAB+ EEP-3A each, %s" % (B, F, what)}
    try:
        check = sorted({0, 1, B // 2, B - 1} & set(range(B)))
        logs = {e: dict(fib=[], ok=[], msc=[[] for _ in PARITY_SUBCH]) for e in check}
        for W in range(n_warm):
            dev.process(F); sf = dev.superframes_stats()
            if W * F < 64:
                info = dev.frame_info(); fib, ok = dev.fibs(); mscs = [dev.msc(i) for i in PARITY_SUBCH]
                for e in check:
                    valid = [f for f in range(F) if info[e, f]["valid"] == 1]
                    for f in valid:
                        logs[e]["fib"].append(np.array(fib[e, f])); logs[e]["ok"].append(np.array(ok[e, f]))
                    for k in range(len(PARITY_SUBCH)):
                        m, fv = mscs[k]
                        logs[e]["msc"][k].append(m[e, fv[e]:4 * len(valid)].tobytes())
        wf0, passes0, fb0 = dev.wide_sync_stats(); rep0 = dev.replayed_batches(); ss0 = dev.sync_stats()
        sync_(); t0 = time.perf_counter(); acc = {}; n_valid = 0; fib_ok = fibs = 0; sf_acc = np.zeros(4, np.int64)
        for _ in range(steps):
            dev.process(F); sf = dev.superframes_stats(); fib, ok = dev.fibs_host()
            for k, v in dev.stage_times().items():
                acc[k] = acc.get(k, 0.0) + v
            ok = np.asarray(ok); fib_ok += int(ok.sum()); fibs += ok.size; sf_acc += sf.sum(0)
            n_valid += int((dev.frame_info()["valid"] == 1).sum())
        sync_(); dt = (time.perf_counter() - t0) / steps
        wf1, passes1, fb1 = dev.wide_sync_stats(); rep1 = dev.replayed_batches(); ss1 = dev.sync_stats()
        assert n_valid >= int(0.98 * steps * B * F), "frames demodulated in the timed steps: %d of %d" % (n_valid, steps * B * F)
        if kind == "drift":
            assert fib_ok == fibs, "FIB CRC failures in the drifting signal: %d of %d pass" % (fib_ok, fibs)
        out.update(value=n_valid / steps * FRAME_S / dt, unit="x real-time", ms_per_step=dt * 1e3, frames_per_step=n_valid / steps,
                   stages_ms={k: v / steps for k, v in acc.items()}, demod_ms=acc.get("demod", 0.0) / steps, msc_viterbi_ms=acc.get("msc_viterbi", 0.0) / steps,
                   wide_sync_stats={"frames_accepted_from_the_wide_pass": int((wf1 - wf0).sum()), "frames": n_valid, "passes": int(passes1 - passes0), "passes_that_needed_the_serial_chain": int(fb1 - fb0)},
                   replayed_batches=int(rep1 - rep0), failed_window_searches=int(ss1[0].sum() - ss0[0].sum()), frames_settled_by_ordered_float_sums=int(ss1[1].sum() - ss0[1].sum()),
                   fib_crc_ok=[fib_ok, fibs],
                   superframes={"synchronised": int(sf_acc[0]), "rs_corrected_symbols": int(sf_acc[1]), "rs_uncorrectable": int(sf_acc[2]), "au_crc_failures": int(sf_acc[3])},
                   per_ensemble={k: [float(np.min(np.abs(v))), float(np.max(np.abs(v)))] for k, v in par.items() if v is not None and k in ("ppm", "snr_db")})
        with tempfile.TemporaryDirectory() as td:
            recs = []
            for e in check:
                path = os.path.join(td, "rec%d.npy" % e); np.save(path, iq[e, :(n_warm * F + 8) * T_F].cpu().numpy() if not loop else iq[e].cpu().numpy()); recs.append(path)
            env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
            _run_receivers(recs, len(recs), 1 if not loop else max(2, -(-80 // rec_frames)), "port", env, td)
            n, m = _compare_with_receivers(td, "port", check, logs)
        out.update(parity=True, parity_detail="FIBs + CRC flags of %d frames and %d MSC bytes of each of sub-channels %s of ensembles %s equal the oracle's (C restatement, same rows)" % (n, m, list(PARITY_SUBCH), check))
    except AssertionError as ex:
        out.update(parity=False, parity_error=str(ex))
    finally:
        dev.close()
    if kind == "drift" and device == "cuda" and sched in (1, 3):
        # the same signal with the next batch's synchroniser always queued BEHIND the decoder (dabphy_config.sync_early = 1; the default
        # is in front): timing only
        try:
            dev = workload.open_receiver(capi, lib_path, iq, F, subchs, device=local, pipeline_sync=sched, loop=loop, sync_early=1)
            for _ in range(n_warm):
                dev.process(F); dev.superframes_stats()
            sync_(); t0 = time.perf_counter()
            for _ in range(steps):
                dev.process(F); dev.superframes_stats(); dev.fibs_host()
            sync_(); dt = (time.perf_counter() - t0) / steps
            out["synchroniser_always_behind_the_decoder"] = {"ms_per_step": dt * 1e3, "value": B * F * FRAME_S / dt, "stages_ms": dev.stage_times()}
            dev.close()
        except Exception as ex:
            out["synchroniser_always_behind_the_decoder"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    return out


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


_BUILD_ID = {}


def _build_id(lib_path):
    """hashes of the sources and of the library file actually loaded (welle_io_amd/buildid.py), computed once"""
    if not _BUILD_ID:
        from welle_io_amd import buildid
        _BUILD_ID.update(src_sha256=buildid.source_sha256(), lib_sha256=buildid.file_sha256(lib_path))
    return _BUILD_ID


def _static_profile(name, B, F, lib_path, stale):
    """counter figures collected by tools/make_profiles.sh in their own rocprofv3 passes (NOT measured by this run; keyed by batch
    geometry AND by the build: a profile whose src_sha256 is not that of the sources this library was built from is not reported --
    its name is appended to `stale` instead)"""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    try:
        pj = json.load(open(path))
    except Exception:
        return None
    if "ensembles" in pj and not (pj.get("ensembles") == B and pj.get("frames") == F):
        return None
    if pj.get("src_sha256") != _build_id(lib_path)["src_sha256"]:
        stale.append("profiles/%s (src_sha256 %s..., this build %s...)" % (name, str(pj.get("src_sha256"))[:12], _build_id(lib_path)["src_sha256"][:12]))
        return None
    return pj


def _extra(cmd, env_extra, timeout):
    """one of the side measurements (tools/): own process, own handle, after the timed region; returns its JSON line or an error record"""
    try:
        r = subprocess.run([sys.executable] + cmd, env=dict(os.environ, **env_extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        return json.loads(lines[-1]) if lines else {"error": "rc %d: %s" % (r.returncode, r.stderr[-300:])}
    except Exception as ex:
        return {"error": "%s: %s" % (type(ex).__name__, ex)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--ensembles", type=int, default=256, help="ensembles per GPU")
    ap.add_argument("--frames", type=int, default=32, help="transmission frames per ensemble and step (round 1 and most of round 2 ran 20; DESIGN.md section 6 has the sweep)")
    ap.add_argument("--cfo-max-hz", type=float, default=60.0, help="per-ensemble carrier frequency offsets are drawn from +-this (small enough for DQPSK to decode from the first frame on, so every ensemble keeps the same frame count; the oscillator cost does not depend on the value)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baseline (and with it the parity check of this run)")
    ap.add_argument("--no-alt-schedule", action="store_true", help="skip the extra (untimed) pass with the other pipelined schedule")
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements behind the timed region (heterogeneous multiplex, single-ensemble facade latency, host-u8 PCIe-inclusive rate)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under a launcher: start the N ranks ourselves, one process per GPU; rank 0's JSON line is this process's output
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    # TEST ONLY (tests/test_dist_gloo.py, GPU-less machine): the N-rank GLUE of this file -- argument handling, rendezvous, per-step gather,
    # max-over-ranks timing, every rank's own parity leg, the parked ranks, the line -- over gloo with every rank's handle on the kernels'
    # CPU execution model (tests/hipemu).  Nothing of it is a measurement; the line says "data": "emulator".
    emu = os.environ.get("DABPHY_BENCH_EMU") == "1"
    assert emu or torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    assert world == args.gpus or "WORLD_SIZE" in os.environ, (world, args.gpus)
    if "WORLD_SIZE" in os.environ and world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d ranks\n" % (args.gpus, world)); sys.exit(2)
    if os.environ.get("DABPHY_SHARE_GPU") == "1":            # TEST ONLY (1-GPU box): several ranks on one device; RCCL refuses that, so the
        local = local % torch.cuda.device_count()            # collective then runs over gloo (DABPHY_DIST_BACKEND=gloo)
    cdev = "cpu" if emu else "cuda"
    dev_sync = (lambda: None) if emu else torch.cuda.synchronize
    if not emu:
        torch.cuda.set_device(local)
    dist = None; backend = None
    if world > 1 or os.environ.get("DABPHY_FORCE_DIST") == "1":    # (the env switch lets a 1-GPU box exercise the RCCL path with one rank)
        import torch.distributed as dist
        backend = "gloo" if emu else os.environ.get("DABPHY_DIST_BACKEND", "nccl")
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
        # a rendezvous or an RCCL bring-up that hangs (a missing rank, an IPC handle the driver refuses) would otherwise sit silently
        # until the driver's own limit: fail fast and say where
        import datetime
        import threading

        def _hung():
            sys.stderr.write("bench.py rank %d/%d: torch.distributed initialisation (backend %s, master %s:%s) did not finish within 240 s -- giving up\n"
                             % (rank, world, backend, os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT")))
            sys.stderr.flush(); os._exit(3)
        dog = threading.Timer(240.0, _hung); dog.daemon = True; dog.start()
        dist.init_process_group(backend, timeout=datetime.timedelta(seconds=600))   # "nccl" = RCCL (no device_id: eager RCCL initialisation prints to stdout, and stdout is one JSON line)
        if backend == "nccl":                    # the first collective is where RCCL really connects the ranks: do it under the watchdog
            t_ = torch.zeros(1, device="cuda"); dist.all_reduce(t_); torch.cuda.synchronize()
        dog.cancel()
    load_package()
    from welle_io_amd import capi, workload
    from welle_io_amd.distributed import FibGatherer

    B, F = args.ensembles, args.frames
    rec_frames = workload.rec_frames_for(F)                # the looping recording is at least one batch long: a step reads every sample once
    # (every rank builds its own recordings on its host cores before the first barrier: bounded like the rendezvous, so that a rank that
    # never arrives says so instead of leaving the others in a collective until its timeout)
    import threading

    def _slow_signal():
        sys.stderr.write("bench.py rank %d/%d: building the synthetic signal took more than 900 s -- giving up\n" % (rank, world)); sys.stderr.flush(); os._exit(4)
    dog2 = threading.Timer(900.0, _slow_signal); dog2.daemon = True; dog2.start()
    iq, cfo_hz, base, txs = workload.make_batch(B, rank=rank, cfo_max_hz=args.cfo_max_hz, device=cdev, rec_frames=rec_frames, n_distinct=min(4, B))
    dog2.cancel()
    N = iq.shape[1]
    dev_sync()
    subchs = txs[0].subchs
    lib_path = os.environ.get("DABPHY_LIB", os.path.join(ROOT, "tests", "hipemu", "libdabphy_emu.so") if emu else os.path.join(PKG_DIR, "libdabphy_hip.so"))
    chunk = int(os.environ.get("DABPHY_DEMOD_CHUNK", "0"))
    sched = int(os.environ.get("DABPHY_PIPELINE", "1"))
    dev = workload.open_receiver(capi, lib_path, iq, F, subchs, device=0 if emu else local, pipeline_sync=sched, demod_chunk=chunk)

    # the one collective of the path: send buffer, rank 0's receive list and its page-locked landing area exist before the first step
    gatherer = FibGatherer(dist, rank, world, (B, F, 12, 32), (B, F, 12), ("cuda:%d" % local) if backend == "nccl" else "cpu") if dist is not None else None

    def step():
        dev.process(F)
        sf = dev.superframes_stats()                       # DAB+ superframe filter of all 18 sub-channels on the device: Fire-code sync, RS, AU CRCs
        if dist is not None:                               # final FIC gather to rank 0 over RCCL/xGMI, straight from the library's HBM buffers
            if backend == "nccl":
                d_fib, d_ok = dev.fibs_device()
            else:                                          # (gloo, test only: the library's page-locked host copies)
                d_fib, d_ok = dev.fibs_host()
            got = gatherer.gather(d_fib, d_ok)
            fib, ok = got[0] if rank == 0 else (None, None)     # rank 0 now holds every rank's FIBs on its host
        else:
            fib, ok = dev.fibs_host()                      # decoded FIBs + CRC flags on the host (page-locked copies that came back with the batch)
        return fib, ok, sf

    # the device idles at a fraction of its clock while the signal is built: bring it up before anything is decoded, so that the
    # warm-up steps (and a profiler's per-kernel averages over the whole run) see the clocks the timed steps see
    if not emu:
        spin = torch.randn(4096, 4096, device="cuda")
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < 0.25:
            spin = torch.tanh(spin @ spin)
            torch.cuda.synchronize()
        del spin
    # warm-up: acquisition, time-de-interleaver fill, superframe synchronisation.  Rank 0 logs what two ensembles of the batch (first
    # and last) deliver from the very first frame on: the parity leg below compares it with CPU receivers decoding the same rows.
    # Rank 0 of a single-GPU run: eight ensembles spread over the batch (all four recordings, both ends), compared below with the real
    # reference backend AND the oracle.  Multi-rank runs: every rank checks the first and the last ensemble of ITS shard against the
    # oracle (outside the timed region) and the verdicts are all-reduced into the line (parity_check.ranks_ok).
    if world == 1:
        check = sorted({0, 1, 2, 3, B // 2, B - 3, B - 2, B - 1} & set(range(B)))
    else:
        check = sorted({0, B - 1})
    logs = {e: dict(fib=[], ok=[], msc=[[] for _ in PARITY_SUBCH]) for e in check}
    for W in range(max(2, args.warmup)):
        fib, ok, sf = step()
        if check and (world > 1 or not args.no_cpu_baseline) and W * F < 64:           # (the CPU receivers keep their first 64 frames for the comparison)
            info = dev.frame_info()
            mscs = [dev.msc(i) for i in PARITY_SUBCH]
            lfib, lok = (fib, ok) if dist is None else dev.fibs()        # (with ranks, `fib` is the gathered set and only rank 0 has it)
            for e in check:
                valid = [f for f in range(F) if info[e, f]["valid"] == 1]
                for f in valid:
                    logs[e]["fib"].append(np.array(lfib[e, f])); logs[e]["ok"].append(np.array(lok[e, f]))
                for k in range(len(PARITY_SUBCH)):
                    m, fv = mscs[k]
                    logs[e]["msc"][k].append(m[e, fv[e]:4 * len(valid)].tobytes())
    fib, ok, sf = step()
    # sanity outside the timed region: all FIBs pass CRC; every sub-channel of every ensemble delivers its 4F/5 superframes per
    # step, none uncorrectable, every access unit passes its CRC; the FIBs of ensemble 0 are the transmitted ones
    if dist is not None:
        fib, ok = dev.fibs()                               # every rank checks its own shard
    fib = np.asarray(fib); ok = np.asarray(ok)
    if os.environ.get("DABPHY_BENCH_NOCHECK") == "1":      # (kernel timing experiments with deliberately wrong results: tools only)
        sf[:, 0] = len(subchs) * (4 * F // 5); sf[:, 2:] = 0
    assert ok.all(), "FIB CRC failures in the benchmark signal"
    assert (sf[:, 0] >= len(subchs) * (4 * F // 5)).all() and (sf[:, 0] <= len(subchs) * ((4 * F + 4) // 5)).all(), "superframe filter: %s" % sf[:4]
    assert (sf[:, 2] == 0).all() and (sf[:, 3] == 0).all(), "superframe filter: %s" % sf[:4]
    sent = set(b"".join(f) for f in txs[0].fib_log)
    assert all(fib[0, f].tobytes() in sent for f in range(F)), "decoded FIBs differ from the transmitted ones"
    lib_sha_note = _build_id(lib_path)

    stage_acc = {}
    if dist is not None:
        dist.barrier()
    dev_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        for k, v in dev.stage_times().items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v
    dev_sync()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    demod_ms_ranks = None
    if dist is not None:
        tt = torch.tensor([dt], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # the FFT stage's kernel time of every rank (HIP events of each rank's own launches): the roofline is quoted on the slowest
        td_ = torch.zeros(world, device=tt.device, dtype=torch.float64); td_[rank] = stage_acc.get("demod", 0.0) / args.steps
        dist.all_reduce(td_, op=dist.ReduceOp.SUM)
        demod_ms_ranks = [float(v) for v in td_.cpu()]

    # ---- multi-rank runs: every rank proves ITS shard (outside the timed region): the first and the last ensemble of the shard through
    # the oracle on this host, FIBs + CRC flags + MSC bytes of three sub-channels from the very first frame on; the verdicts are summed
    ranks_ok = None; rank_err = None
    if world > 1:
        import tempfile
        flag = 0
        try:
            with tempfile.TemporaryDirectory() as td:
                recs = []
                for e in check:
                    path = os.path.join(td, "rec%d.npy" % e); np.save(path, iq[e].cpu().numpy()); recs.append(path)
                env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
                # every rank's leg runs on the same host: each takes its share of the cores (at least one receiver at a time), so that
                # N ranks never start more receivers together than the host has cores
                cores_ = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
                share = max(1, min(len(recs), cores_ // world))
                for i0 in range(0, len(recs), share):
                    part = list(range(i0, min(len(recs), i0 + share)))
                    tdp = os.path.join(td, "part%d" % i0); os.makedirs(tdp)
                    _run_receivers([recs[i] for i in part], len(part), max(2, -(-80 // rec_frames)), "port", env, tdp)
                    _compare_with_receivers(tdp, "port", [check[i] for i in part], logs)
            flag = 1
        except Exception as ex:
            rank_err = "rank %d: %s: %s" % (rank, type(ex).__name__, ex)
            sys.stderr.write("bench.py parity leg, %s\n" % rank_err)
        tf = torch.tensor([flag], device="cuda" if backend == "nccl" else "cpu", dtype=torch.int64)
        dist.all_reduce(tf, op=dist.ReduceOp.SUM)
        ranks_ok = int(tf.item())

    # Rank 0 measures the CPU baseline on the host's cores after the timed region.  The other ranks must not spin in an RCCL barrier on
    # those cores meanwhile: they sleep on a file rank 0 writes when its line is out (bounded: the watchdog below), and only then does
    # everybody meet in the final barrier.
    park = os.path.join(os.environ.get("TMPDIR", "/tmp"), "dabphy_bench_%s_%s.done" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "run"))) if world > 1 else None
    if park and rank == 0 and os.path.exists(park):
        os.remove(park)
    if dist is not None:
        dist.barrier()                                     # (the stale file of an earlier run is gone before anybody looks for it)
    if park and rank != 0:
        t_park = time.time()
        while not os.path.exists(park):
            if time.time() - t_park > 1500:
                sys.stderr.write("bench.py rank %d/%d: rank 0 did not finish its CPU baseline within 1500 s -- giving up\n" % (rank, world)); sys.stderr.flush(); os._exit(5)
            time.sleep(0.25)
    if rank == 0:
        n_simd = 1024 if emu else 4 * torch.cuda.get_device_properties(local).multi_processor_count
        ms_step = dt / args.steps * 1e3
        value = world * B * F * FRAME_S / (dt / args.steps)
        stages = {k: v / args.steps for k, v in stage_acc.items()}
        demod_ms = max(demod_ms_ranks) if demod_ms_ranks else stages["demod"]
        ach = B * F * ALG_BYTES_DEMOD_PER_FRAME / (demod_ms * 1e-3) / 1e9
        stale = []
        pj = _static_profile("demod_hbm_traffic.json", B, F, lib_path, stale)
        n_cw_steps = B * F * (4 * 774 + 72 * 1542)
        line = {
            "metric": "DAB Mode-I ensembles/s (x real-time)", "value": value, "unit": "x real-time (ensembles decoded concurrently)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 (FFT/demap) + u16 (Viterbi metrics) + u8 (GF(256))", "data": "emulator: NOT a measurement (DABPHY_BENCH_EMU=1: the N-rank glue on the kernels' CPU execution model)" if emu else "synthetic",
            "config": {"workload": "1xMI355X: batch of %d synthetic Mode-I ensembles x %d frames (2.048 Msps cf32, HBM-resident, 18 x 64 kbit/s DAB+ EEP-3A sub-channels each), full chain incl. Viterbi + Reed-Solomon" % (B, F),
                       "ensembles_per_gpu": B, "frames_per_step": F, "recording_frames": rec_frames, "cfo_hz": "uniform +-%g per ensemble" % args.cfo_max_hz, "frames_per_s": world * B * F / (dt / args.steps), "sharding": "by ensemble, %d per GPU" % B,
                       "demod_chunk": dev.demod_chunk(), "exact_batch_mode": "on (dabphy_config.no_batch_replay = 0): batches decoded a second time in this run: %d" % dev.replayed_batches(), "superframe_wide_pass": "(ensemble, sub-channel) batches settled by the filter's wide pass / tried: %d / %d (the rest walked frame by frame)" % dev.wide_superframe_stats(), "parity_test": "tests/test_gpu_bench_config.py decodes this configuration against the oracle"},
            "rccl_ranks": world if (dist is not None and backend == "nccl") else 0,
            "roofline": {"kernel": "k_demod (NCO + 2048-pt FFT + DQPSK demap + freq de-interleave)", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": pj.get("hbm_bytes_per_launch") if pj else None,
                         "traffic_source": ("static: profiles/demod_hbm_traffic.json (FETCH_SIZE / WRITE_SIZE passes of tools/make_profiles.sh on this command; not measured by this run)" if pj else None),
                         "algorithmic_bytes_per_launch": B * F * ALG_BYTES_DEMOD_PER_FRAME, "kernel_ms": demod_ms,
                         "kernel_ms_per_rank": demod_ms_ranks, "kernel_ms_note": ("per launch, HIP events on the launch's stream; with ranks: the SLOWEST rank's (each rank launches the kernel on its own %d x %d shard)" % (B, F)) if demod_ms_ranks else "per launch, HIP events on the launch's stream",
                         "bound_note": "the HBM roofline is what BASELINE.json's metric asks this stage to be measured against; the kernel itself spends about three quarters of its time on arithmetic, LDS traffic and barriers (642 VALU instructions per thread and symbol of bit-exact KISS butterflies, demapper and double-precision oscillator: DESIGN.md 4.1), at the package power limit",
                         "classic_c2c_GBps": B * F * ALG_BYTES_FFT_CLASSIC_PER_FRAME / (demod_ms * 1e-3) / 1e9,
                         "survey_8d_fused_GBps": B * F * (196608 * 8 + 75 * 3072) / (demod_ms * 1e-3) / 1e9},
            "stages_ms": stages,
        }
        # ---- Viterbi stage: VALU issue roofline (the kernel holds its path metrics in VGPRs; no LDS traffic to be efficient with)
        vit_ms = stages.get("msc_viterbi", 0.0)
        vj = _static_profile("viterbi_counters.json", B, F, lib_path, stale)
        ub = None
        try:
            ub = json.load(open(os.path.join(ROOT, "profiles", "valu_rate.json")))
        except Exception:
            ub = None
        rv = {"kernel": "k_viterbi_fused: all 18 sub-channels and the FIC in one launch (lane = codeword, 64 states in 32 VGPRs of u16 pairs; SWAR additions in a six-layout rotating state pairing, viterbi_acs.h; de-interleave + depuncture gather fused in through an LDS window ring; work list pulled by persistent waves)", "bound": "valu",
              "kernel_ms": vit_ms, "kernel_ms_note": "HIP events around the launch inside the pipelined step: the next batch's synchroniser and the SNR sums run beside it",
              "codeword_steps_per_s": n_cw_steps / (vit_ms * 1e-3) if vit_ms > 0 else None,
              "algorithmic_bytes": B * F * (72 * (4 * 1542 + 1536 // 8) + 4 * (4 * 774 + 768 // 8)), "hbm_bytes": vj.get("hbm_bytes_per_launch") if vj else None,
              "hbm_bytes_if_narrow_requests_are_tallied_in_full": vj.get("hbm_bytes_if_narrow_requests_are_tallied_in_full") if vj else None,
              "decision_bytes_written_plus_read": vj.get("decision_bytes_written_plus_read") if vj else None,
              "hbm_note": "the decision array (8 bytes per trellis step and code word, written by the forward pass, read back by the traceback) is 3.9 x the algorithmic bytes on its own: what bounds this kernel now (DESIGN.md 4.2)",
              "valu_insts_per_launch": vj.get("valu_insts_per_launch") if vj else None,
              "counter_source": "static: profiles/viterbi_counters.json (SQ_INSTS_VALU / FETCH_SIZE / WRITE_SIZE passes of tools/make_profiles.sh; not measured by this run)" if vj else None}
        if vj and vj.get("valu_insts_per_launch") and vit_ms > 0:
            ips = vj["valu_insts_per_launch"] / (vit_ms * 1e-3)
            rv.update(unit="wave-instructions/s", achieved=ips, peak=n_simd * CLOCK_HZ / 2.0, frac=ips / (n_simd * CLOCK_HZ / 2.0),
                      peak_note="MI355X_MICROARCH.md: one plain wave64 VALU instruction per SIMD every 2 cycles at 2.4 GHz")
            if ub:
                cyc = ub["cycles_per_instruction_kernel_mix"]
                rv.update(peak_measured_mix=n_simd * CLOCK_HZ / cyc, frac_of_measured_mix=ips / (n_simd * CLOCK_HZ / cyc),
                          measured_mix_note="profiles/valu_rate.json: issue cost of this kernel's instruction mix from tools/ubench/valu_rate.hip (packed 16-bit ops %.2f cycles, plain %.2f)" % (ub["cycles_packed"], ub["cycles_plain"]))
        if vj and vj.get("lds_idx_active_cycles"):
            # north star: "LDS-bank efficiency for Viterbi ACS".  The ACS itself touches no LDS (path metrics are register-resident); the
            # kernel's LDS traffic is the byte gather from its de-interleaver window ring
            rv.update(lds_bank_efficiency=1.0 - vj["lds_bank_conflict_cycles"] / vj["lds_idx_active_cycles"],
                      lds_insts_per_trellis_step=vj["lds_insts_per_launch"] / (n_cw_steps / 64.0) if vj.get("lds_insts_per_launch") else None,
                      lds_note="1 - SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the launch (static, profiles/viterbi_counters.json): the window ring's rows are 20 bytes (five dwords) apart, so the byte reads of consecutive rows fall on different banks (round 2, 16-byte pitch: 0.28); the add-compare-select itself runs in VGPRs (DESIGN.md 4.2)")
        line["roofline_viterbi"] = rv
        # which build this line was measured on, and which committed counter profiles were NOT reported because they belong to another one
        line["profile_build"] = dict(lib_sha_note, stale_profile=stale or None,
                                     note="static counter figures (roofline.traffic, roofline_viterbi.hbm_bytes / valu_insts / lds_*) are reported only when the profile's src_sha256 equals this build's (tools/make_profiles.sh + tools/collect_profiles.py regenerate them)")
        # what a plain device-to-device copy moves on this box (read + write), for scale next to the 8 TB/s specification the fraction is
        # taken against ("measure the denominator"); never used as `peak`.  The library's own float4 grid-stride copy
        # (dabphy_time_copy, the recipe MI355X_MICROARCH.md quotes 6.29 TB/s for; tools/ubench/copy_f4.hip sweeps it), 2 GiB in + 2 GiB out
        # per pass; torch's copy_ beside it (round 3's denominator: it flattered)
        try:
            if emu:
                raise RuntimeError("emulator run: no copy measured")
            sweep = {k: dev.time_copy(2 << 30, k, 5) for k in (2, 4, 8, 16)}
            cg = max(sweep.values())
            line["roofline"]["measured_copy_GBps"] = cg
            line["roofline"]["frac_of_achievable"] = ach / cg
            line["roofline"]["measured_copy_note"] = "float4 grid-stride device copy of 2 GiB (read + write counted), best of 2 / 4 / 8 / 16 work-groups per CU (%s), after this run's timed region on the same device (dabphy_time_copy; tools/ubench/copy_f4.hip is the full sweep, profiles/r04_copy_f4.txt)" % ", ".join("%d: %.0f" % kv for kv in sorted(sweep.items()))
            src = iq[: max(1, B // 2)]; dst = torch.empty_like(src)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            dst.copy_(src); torch.cuda.synchronize()
            e0.record()
            for _ in range(3):
                dst.copy_(src)
            e1.record(); torch.cuda.synchronize()
            line["roofline"]["torch_copy_GBps"] = 3 * 2 * src.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del dst
        except Exception as ex:
            line["roofline"]["measured_copy_error"] = "%s: %s" % (type(ex).__name__, ex)
        if world == 1 and not args.no_extras:
            try:
                line["msc_drain"] = msc_drain_leg(dev, torch, B, F, args.steps, B * len(subchs))
                line["msc_drain"]["vs_headline"] = line["msc_drain"]["value"] / value
            except Exception as ex:
                line["msc_drain"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        line["config"]["schedule"] = {0: "serial synchroniser", 1: "pipelined: the next batch's synchroniser starts behind this batch's demod kernel",
                                      2: "pipelined: the next batch's synchroniser starts at once (shares the device with the demod kernel)",
                                      3: "pipelined two batches ahead: the synchroniser of batch k + 2 starts behind batch k's demod kernel"}[sched]
        if world == 1 and sched in (1, 2, 3) and not args.no_alt_schedule:
            # the other pipelined schedules, measured the same way right after the timed region (reported, never `value`)
            dev.close(); dev = None
            line["alt_schedules"] = []
            for alt in [m for m in (1, 2, 3) if m != sched]:
                dev2 = workload.open_receiver(capi, lib_path, iq, F, subchs, device=local, pipeline_sync=alt, demod_chunk=chunk)

                def step2():
                    dev2.process(F); dev2.superframes_stats(); return dev2.fibs()
                for _ in range(4):
                    step2()
                torch.cuda.synchronize(); t1 = time.perf_counter(); dm = 0.0
                for _ in range(args.steps):
                    step2(); dm += dev2.stage_times()["demod"]
                torch.cuda.synchronize(); dt2 = (time.perf_counter() - t1) / args.steps
                dm /= args.steps
                line["alt_schedules"].append({"pipeline_sync": alt, "value": B * F * FRAME_S / dt2, "ms_per_step": dt2 * 1e3, "demod_kernel_ms": dm,
                                              "roofline_frac": B * F * ALG_BYTES_DEMOD_PER_FRAME / (dm * 1e-3) / 1e9 / HBM_PEAK_GBPS})
                dev2.close()
            line["alt_schedule"] = line["alt_schedules"][0]
        if not args.no_cpu_baseline:
            # rank 0's host cores, after the timed region, whatever the number of ranks (the other ranks wait at the final barrier):
            # the same bounded sample, over the rows rank 0 logged -- its parity leg against the real reference backend and the oracle
            if dev is not None and world > 1:
                pass                                     # (the handle stays open: nothing of the baseline runs on the device)
            rows = {e: iq[e].cpu().numpy() for e in check}
            # (DABPHY_BENCH_QUICK=1, tests/test_gpu_bench_entry.py only: half the sample -- the test checks the line's contract, not its figures)
            line["cpu_baseline"], pc = cpu_baseline(rows, n_loops=max(1, (120 if os.environ.get("DABPHY_BENCH_QUICK") == "1" else 240) // rec_frames), gpu_logs=logs)
            if world == 1:
                line["parity_check"] = pc
                line["parity_check"].update(ranks_ok=1, ranks=1)
        if world > 1:
            line["parity_check"] = {"ranks_ok": ranks_ok, "ranks": world, "per_rank": "ensembles %s of the rank's own shard: FIBs + CRC flags + MSC bytes of sub-channels %s from the first frame on vs the oracle (C restatement) on the same rows, outside the timed region" % (check, list(PARITY_SUBCH)),
                                    "fib_equal": ranks_ok == world, "msc_equal": ranks_ok == world, "rank0_error": rank_err}
            if not args.no_cpu_baseline:
                line["parity_check"]["rank0_against"] = pc.get("against")
        if world == 1 and not args.no_extras:
            # BASELINE configs 2-3 and the PCIe-inclusive path, measured in this run (own processes, own handles, after the timed region;
            # never `value`): the drop-in facade over one ensemble, one frame per call, every getter copied out, the reference's
            # FIBProcessor on the host; and u8 IQ from page-locked host memory, double-buffered over PCIe, at the batch geometry
            if dev is not None:
                dev.close(); dev = None
            del iq; torch.cuda.empty_cache()
            # a multiplex as they are on air instead of 18 identical sub-channels: same batch geometry, own parity leg
            try:
                line["extras"] = {"hetero": hetero_leg(capi, workload, torch, lib_path, B, F, args.steps, local, sched)}
            except Exception as ex:
                line["extras"] = {"hetero": {"error": "%s: %s" % (type(ex).__name__, ex)}}
            torch.cuda.empty_cache()
            # a batch of independent ensembles: five multiplexes, five selections, interleaved over the batch
            try:
                line["extras"]["mixed_layouts"] = mixed_layouts_leg(capi, workload, torch, lib_path, B, F, args.steps, local, sched)
            except Exception as ex:
                line["extras"]["mixed_layouts"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
            torch.cuda.empty_cache()
            # the headline's geometry on signals as a receiver meets them: drifting sample clocks, low SNR (own parity legs)
            for kind in ("drift", "low_snr"):
                try:
                    line["extras"][kind] = channel_leg(capi, workload, torch, lib_path, B, F, args.steps, local, sched, kind)
                except Exception as ex:
                    line["extras"][kind] = {"error": "%s: %s" % (type(ex).__name__, ex)}
                torch.cuda.empty_cache()
            # the latency regime: ONE ensemble, 1 / 4 / 8 / 16 frames per call, both Viterbi kernels (the default picks the state-parallel one here)
            if os.environ.get("DABPHY_BENCH_QUICK") != "1":      # (tests/test_gpu_bench_entry.py skips this sweep: tests/test_gpu_stream.py covers the kernels it times)
                line["extras"]["short_batches"] = _extra([os.path.join(ROOT, "tools", "sweep_decode_shape.py"), "--json"], {}, 300)
            line["facade"] = _extra([os.path.join(ROOT, "tools", "bench_facade.py"), "--json"], {}, 420)
            line["host_u8"] = _extra([os.path.join(ROOT, "tools", "bench_host_u8.py")], {"HOSTU8_B": str(B), "HOSTU8_F": str(F), "HOSTU8_STEPS": "3"}, 300)
        print(json.dumps(line), flush=True)
        if park:
            open(park, "w").write("done\n")
    if dev is not None:
        dev.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if park and rank == 0 and os.path.exists(park):
        os.remove(park)


if __name__ == "__main__":
    main()
