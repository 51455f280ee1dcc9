"""N > 1 path on CPU: 2 ranks (torch.distributed, gloo, 127.0.0.1), each decodes its own shard of ensembles with the
kernels in the tests/hipemu execution model, FIBs are gathered to rank 0 exactly as bench.py does over RCCL, and rank 0
checks every ensemble against the oracle."""
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import conftest, refapi as R
    from welle_io_amd import capi, synth
    from welle_io_amd.distributed import gather_fibs, shard_range
    rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    N_ENS, F = 4, 3
    lo, hi = shard_range(N_ENS, rank, world)
    streams = [synth.make_stream(5, snr_db=15, seed=40 + e, eid=0x2000 + e) for e in range(N_ENS)]
    d = capi.DabPhy(n_ensembles=hi - lo, max_frames=F, lib_path=conftest.EMU_LIB, disable_coarse=True)
    d.stream_upload(np.stack(streams[lo:hi]))
    d.process(F)
    fib, ok = d.fibs()
    res = gather_fibs(dist, fib, ok, rank, world)
    if rank == 0:
        assert len(res) == world
        e = 0
        for fib_r, ok_r in res:
            for b in range(fib_r.shape[0]):
                o = R.orc_receiver_run(streams[e], disable_coarse=True)
                ref = o["fib"][:12 * F].reshape(F, 12, 33)
                assert np.array_equal(ok_r[b], ref[:, :, 0]), e
                assert np.array_equal(fib_r[b], ref[:, :, 1:]), e
                assert fib_r[b][0].tobytes() != res[0][0][0][0].tobytes() or e == 0     # ensembles really differ
                e += 1
        assert e == N_ENS
        print("DIST_OK")
    dist.barrier()
    dist.destroy_process_group()
''')


def test_two_ranks_shard_and_gather(emu, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\\n".join(outs)
    assert "DIST_OK" in outs[0]


def test_eight_ranks_of_bench_py_over_gloo(emu):
    """`bench.py --gpus 8` as the driver launches it (torch.distributed.run, 8 ranks, 127.0.0.1) with every rank's handle on the kernels'
    CPU execution model (DABPHY_BENCH_EMU=1, TEST ONLY) and the collective over gloo: the glue of the N > 1 line -- argument handling,
    rendezvous, the preallocated per-step gather (distributed.FibGatherer), max-over-ranks timing, every rank's own parity leg against
    the oracle, the ranks parked while rank 0 writes the line -- cannot be what fails the first real 8-GPU run.  2 ensembles x 2 frames
    per rank."""
    import json
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, DABPHY_BENCH_EMU="1", OMP_NUM_THREADS="1", DABPHY_PIPELINE="1", DABPHY_SP_MAX_CW="0")      # (the execution model runs the state-parallel kernels a wavefront at a time: lane per code word here, as in tests/conftest.py)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "7", "--ensembles", "2", "--frames", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["steps"] == 2 and j["scaling"] == "weak" and j["data"].startswith("emulator")
    assert j["parity_check"]["ranks_ok"] == 8 and j["parity_check"]["ranks"] == 8, j["parity_check"]
    assert len(j["roofline"]["kernel_ms_per_rank"]) == 8 and all(v > 0 for v in j["roofline"]["kernel_ms_per_rank"])
    assert abs(j["value"] - 8 * 2 * 2 * 0.096 / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]
    # a launcher that started another number of ranks than --gpus names is refused, not mis-reported
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--no-cpu-baseline"], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert bad.returncode == 2 and "--gpus 4" in bad.stderr, (bad.returncode, bad.stderr[-500:])
