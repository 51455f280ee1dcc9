#!/usr/bin/env python
"""bench.py -- DAB Mode-I PHY hot path on MI355X: whole-job throughput, FFT-stage HBM roofline, CPU baseline.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.  For N > 1 the
driver starts one process per GPU with torch.distributed.run; every rank decodes its own block of ensembles (the
path shards by ensemble, no data-path collective) and the decoded FIBs are gathered to rank 0 over RCCL per step.

A "step" is one pass of the whole hot path (time/frequency sync, NCO, 2048-pt FFT, DQPSK demap, de-interleavers,
depuncture, Viterbi, energy dispersal, FIB CRC, Reed-Solomon) over one batch of B ensembles x F transmission frames
of synthetic 2.048 Msps complex-float IQ that is already resident in HBM (BASELINE.json config
"1xMI355X: batch of 256 synthetic Mode-I ensembles").  value = ensembles decoded in real time (x real-time) summed
over all GPUs = N*B*F*0.096 s / step time.
"""
import argparse
import json
import os
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


def make_base_streams(n_distinct, n_frames, seed0):
    """n_distinct looping recordings of n_frames frames each: canonical ensemble (18 x 64 kbit/s DAB+ EEP-3A), RS-valid
    payload periodic in 4*n_frames CIFs, transmitter run for one extra period first so the loop point is seamless"""
    from welle_io_amd import synth
    out, txs = [], []
    for e in range(n_distinct):
        tx = synth.EnsembleTx(eid=0x1000 + seed0 + e, seed=seed0 + e, payload_fn=synth.dabplus_payload_fn(4 * n_frames, seed0 + e))
        for _ in range(n_frames):
            tx.next_frame()
        frames = [tx.next_frame() for _ in range(n_frames)]
        out.append(np.concatenate(frames).astype(np.complex64))
        txs.append(tx)
    return np.stack(out), txs


def _run_receivers(rec, n_proc, n_loops, mode, env):
    """n_proc concurrent receiver processes over the same recording; returns their result records"""
    import subprocess
    args = [sys.executable, os.path.join(ROOT, "tests", "cpu_baseline_worker.py"), rec, str(n_loops)]
    procs = [subprocess.Popen(args + [str(7 + i)] + (["reference"] if mode == "reference" else []),
                              stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env) for i in range(n_proc)]
    for p in procs:
        if p.stdout.readline().strip() != "READY":
            raise RuntimeError("CPU baseline worker did not start")
    for p in procs:
        p.stdin.write("go\n"); p.stdin.flush()            # all receivers start together
    res = [json.loads(p.stdout.readline()) for p in procs]
    for p in procs:
        p.wait()
    return res


def cpu_baseline(base_stream, n_loops):
    """The CPU side of the same workload on this host, a bounded sample (one canonical ensemble with all 18 sub-channels per
    receiver; the reference's own concurrency model is one receiver per ensemble):
      * kind "reference": the REAL reference backend (oracle/_ref, built from /root/reference where that exists) -- RadioReceiver
        with its own 2-3 threads and SSE Viterbi; it also runs the layers above the PHY (FIG parsing, superframe filter, AAC).
        cores/2 receivers run concurrently so that every core is busy.
      * the oracle (C restatement of the PHY only, single-threaded), one receiver per core, as a second figure (or as the
        baseline, kind "port", where oracle/_ref is absent)."""
    import tempfile
    cores = max(1, min(32, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    with tempfile.TemporaryDirectory() as td:
        rec = os.path.join(td, "rec.npy"); np.save(rec, base_stream)
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        res = _run_receivers(rec, cores, n_loops, "port", env)
        ref = None
        if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libwelle_ref.so")):
            try:
                ref = _run_receivers(rec, max(1, cores // 2), max(1, n_loops // 2), "reference", env)
            except Exception:
                ref = None
    port_value = sum(r["frames"] for r in res) * FRAME_S / max(r["seconds"] for r in res)
    port = dict(value=port_value, per_core=res[0]["frames"] * FRAME_S / res[0]["seconds"], receivers=cores,
                fib_ok=sum(r["fib_ok"] for r in res), fibs=sum(r["fibs"] for r in res))
    if ref:
        slowest = max(r["seconds"] for r in ref)
        return dict(value=sum(r["frames"] for r in ref) * FRAME_S / slowest, unit="x real-time (ensembles decoded concurrently by all cores)", cores=cores,
                    kind="reference", per_receiver=ref[0]["frames"] * FRAME_S / ref[0]["seconds"],
                    sample="%d concurrent RadioReceivers of the reference backend (each 2-3 threads; PHY + FIG parsing + superframe filter + AAC; lock-step input so that no frame is dropped) x %d frames "
                           "(%.1f s of IQ each) of the canonical ensemble, 18 sub-channels, slowest receiver %.1f s"
                           % (len(ref), ref[0]["frames"], ref[0]["frames"] * FRAME_S, slowest),
                    fib_ok=sum(r["fib_ok"] for r in ref), fibs=sum(r["fibs"] for r in ref), oracle_port=port)
    return dict(value=port_value, unit="x real-time (ensembles decoded concurrently by all cores)", cores=cores, kind="port",
                per_core=port["per_core"],
                sample="%d receivers x %d frames (%.1f s of IQ each) of the canonical ensemble, 18 sub-channels, oracle C restatement, slowest receiver %.1f s"
                       % (cores, res[0]["frames"], res[0]["frames"] * FRAME_S, max(r["seconds"] for r in res)),
                fib_ok=port["fib_ok"], fibs=port["fibs"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ensembles", type=int, default=256, help="ensembles per GPU")
    ap.add_argument("--frames", type=int, default=20, help="transmission frames per ensemble and step")
    ap.add_argument("--cfo-max-hz", type=float, default=60.0, help="per-ensemble carrier frequency offsets are drawn from +-this (small enough for DQPSK to decode from the first frame on, so every ensemble keeps the same frame count; the oscillator cost does not depend on the value)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-schedule", action="store_true", help="skip the extra (untimed) pass with the other pipelined schedule")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or os.environ.get("DABPHY_FORCE_DIST") == "1":    # (the env switch lets a 1-GPU box exercise the RCCL path: torchrun --nproc-per-node 1)
        import torch.distributed as dist
        dist.init_process_group("nccl")          # RCCL (no device_id: eager RCCL initialisation prints to stdout, and stdout is one JSON line)
    load_package()
    from welle_io_amd import capi, synth
    from welle_io_amd.distributed import gather_fibs

    B, F = args.ensembles, args.frames
    REC = 20            # frames per looping recording: 80 CIFs = a whole number of superframes (5 CIFs) and interleaver periods (16)
    n_distinct = 4
    base, txs = make_base_streams(n_distinct, REC, seed0=100 * rank)
    N = base.shape[1]
    gbase = torch.from_numpy(base).cuda()
    gen = torch.Generator(device="cuda"); gen.manual_seed(1234 + rank)
    iq = torch.empty((B, N), dtype=torch.complex64, device="cuda")
    # per-ensemble carrier frequency offset (a multiple of RATE/N so that the looping recording stays phase-continuous) and
    # AWGN, sigma = 0.02 per axis (SURVEY 8d throughput setting): every ensemble runs its own coarse/fine correctors
    rs = np.random.RandomState(4321 + rank)
    cfo_hz = np.round(rs.uniform(-args.cfo_max_hz, args.cfo_max_hz, B) * N / 2048000.0) * 2048000.0 / N
    n_idx = torch.arange(N, device="cuda", dtype=torch.float64)
    for b in range(B):
        noise = torch.randn((N, 2), generator=gen, device="cuda", dtype=torch.float32) * 0.02
        rot = torch.polar(torch.ones_like(n_idx), n_idx * (2.0 * np.pi * cfo_hz[b] / 2048000.0)).to(torch.complex64)
        iq[b] = gbase[b % n_distinct] * rot + torch.view_as_complex(noise)
    del gbase, n_idx, rot
    torch.cuda.synchronize()

    sched = int(os.environ.get("DABPHY_PIPELINE", "1"))
    dev = capi.DabPhy(n_ensembles=B, max_frames=F, device=local, lib_path=os.environ.get("DABPHY_LIB", os.path.join(PKG_DIR, "libdabphy_hip.so")),
                      want_constellation=False, want_impulse_response=False, disable_coarse=False, pipeline_sync=sched,
                      demod_chunk=int(os.environ.get("DABPHY_DEMOD_CHUNK", "0")))
    dev.stream_bind_device(iq.data_ptr(), N, N, N, loop=True)
    subchs = txs[0].subchs
    dev.set_subchannels([(s.subch_id, s.start_cu, s.size_cu, dev.protection_eep(s.bitrate, s.profile_b, s.level)) for s in subchs])
    dev.set_profiling(True)
    dev.set_auto_superframes(True)                         # the superframe filter rides in process()'s submission

    def step():
        dev.process(F)
        sf = dev.superframes_stats()                       # DAB+ superframe filter of all 18 sub-channels on the device: Fire-code sync, RS, AU CRCs
        fib, ok = dev.fibs()                               # decoded FIBs + CRC flags to the host of this rank
        if dist is not None:                               # final FIC gather to rank 0 over RCCL/xGMI
            gather_fibs(dist, fib, ok, rank, world, device="cuda")
        return fib, ok, sf

    # warm-up: acquisition, time-de-interleaver fill, superframe synchronisation
    for W in range(max(2, args.warmup)):
        step()
    fib, ok, sf = step()
    # sanity outside the timed region: all FIBs pass CRC; every sub-channel of every ensemble delivers its 4F/5 superframes per
    # step, none uncorrectable, every access unit passes its CRC; the FIBs of ensemble 0 are the transmitted ones
    assert ok.all(), "FIB CRC failures in the benchmark signal"
    assert (sf[:, 0] >= len(subchs) * (4 * F // 5)).all() and (sf[:, 0] <= len(subchs) * ((4 * F + 4) // 5)).all(), "superframe filter: %s" % sf[:4]
    assert (sf[:, 2] == 0).all() and (sf[:, 3] == 0).all(), "superframe filter: %s" % sf[:4]
    sent = set(b"".join(f) for f in txs[0].fib_log)
    assert all(fib[0, f].tobytes() in sent for f in range(F)), "decoded FIBs differ from the transmitted ones"

    stage_acc = {}
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        for k, v in dev.stage_times().items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = world * B * F * FRAME_S / (dt / args.steps)
        stages = {k: v / args.steps for k, v in stage_acc.items()}
        demod_ms = stages["demod"]
        ach = B * F * ALG_BYTES_DEMOD_PER_FRAME / (demod_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "demod_hbm_traffic.json")
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                if pj.get("ensembles") == B and pj.get("frames") == F:
                    traffic = pj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        n_cw_steps = B * F * (4 * 774 + 72 * 1542)
        line = {
            "metric": "DAB Mode-I ensembles/s (x real-time)", "value": value, "unit": "x real-time (ensembles decoded concurrently)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 (FFT/demap) + u16 (Viterbi metrics) + u8 (GF(256))", "data": "synthetic",
            "config": {"workload": "1xMI355X: batch of %d synthetic Mode-I ensembles x %d frames (2.048 Msps cf32, HBM-resident, 18 x 64 kbit/s DAB+ EEP-3A sub-channels each), full chain incl. Viterbi + Reed-Solomon" % (B, F),
                       "ensembles_per_gpu": B, "frames_per_step": F, "cfo_hz": "uniform +-%g per ensemble" % args.cfo_max_hz, "frames_per_s": world * B * F / (dt / args.steps), "sharding": "by ensemble, %d per GPU" % B},
            "roofline": {"kernel": "k_demod (NCO + 2048-pt FFT + DQPSK demap + freq de-interleave)", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": B * F * ALG_BYTES_DEMOD_PER_FRAME, "kernel_ms": demod_ms,
                         "classic_c2c_GBps": B * F * ALG_BYTES_FFT_CLASSIC_PER_FRAME / (demod_ms * 1e-3) / 1e9,
                         "survey_8d_fused_GBps": B * F * (196608 * 8 + 75 * 3072) / (demod_ms * 1e-3) / 1e9},
            "stages_ms": stages,
            "viterbi": {"codeword_steps_per_s": n_cw_steps / ((stages["fic"] + stages["msc_viterbi"]) * 1e-3) if stages["msc_viterbi"] > 0 else None,
                        "bound": "VALU int16 (v_pk_add/min_u16), metrics in VGPRs: no LDS traffic"},
        }
        line["config"]["schedule"] = {0: "serial synchroniser", 1: "pipelined: the next batch's synchroniser starts behind this batch's demod kernel",
                                      2: "pipelined: the next batch's synchroniser starts at once (shares the device with the demod kernel)"}[sched]
        if world == 1 and sched in (1, 2) and not args.no_alt_schedule:
            # the other pipelined schedule, measured the same way right after the timed region (reported, never `value`)
            dev.close(); dev = None
            alt = 3 - sched
            dev2 = capi.DabPhy(n_ensembles=B, max_frames=F, device=local, lib_path=os.environ.get("DABPHY_LIB", os.path.join(PKG_DIR, "libdabphy_hip.so")), want_constellation=False,
                               want_impulse_response=False, disable_coarse=False, pipeline_sync=alt, demod_chunk=int(os.environ.get("DABPHY_DEMOD_CHUNK", "0")))
            dev2.stream_bind_device(iq.data_ptr(), N, N, N, loop=True)
            dev2.set_subchannels([(s.subch_id, s.start_cu, s.size_cu, dev2.protection_eep(s.bitrate, s.profile_b, s.level)) for s in subchs])
            dev2.set_profiling(True); dev2.set_auto_superframes(True)
            def step2():
                dev2.process(F); dev2.superframes_stats(); return dev2.fibs()
            for _ in range(3):
                step2()
            torch.cuda.synchronize(); t1 = time.perf_counter(); dm = 0.0
            for _ in range(args.steps):
                step2(); dm += dev2.stage_times()["demod"]
            torch.cuda.synchronize(); dt2 = (time.perf_counter() - t1) / args.steps
            dm /= args.steps
            line["alt_schedule"] = {"pipeline_sync": alt, "value": B * F * FRAME_S / dt2, "ms_per_step": dt2 * 1e3, "demod_kernel_ms": dm,
                                    "roofline_frac": B * F * ALG_BYTES_DEMOD_PER_FRAME / (dm * 1e-3) / 1e9 / HBM_PEAK_GBPS}
            dev2.close()
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(base[0], n_loops=12)
        print(json.dumps(line))
    if dev is not None:
        dev.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
