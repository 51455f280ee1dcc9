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
