"""The kernel sources compiled for the CPU execution model a SECOND time, configured exactly like the shipped library (`make emu-product`:
DEMOD_WAVES=3, no -DDABPHY_EXPERIMENTS, hence no getenv override anywhere): the stream / mixed-class / bench-configuration checks once more,
so that the GPU-less suite also covers the product's own build switches (every other CPU test runs the experiments build `make emu`)."""
import os
import subprocess

import pytest

import parity_cases as P
from conftest import PKG_DIR, ROOT
from welle_io_amd import capi

LIB = os.path.join(ROOT, "tests", "hipemu", "libdabphy_emu_product.so")


@pytest.fixture(scope="module")
def emu_product():
    subprocess.run(["make", "-j8", "emu-product"], cwd=os.path.join(PKG_DIR, "csrc"), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    return LIB


def factory(**kw):
    return capi.DabPhy(lib_path=LIB, **kw)


def test_environment_is_not_read(emu_product, monkeypatch):
    """the experiments build obeys DABPHY_FUSED_MSC=0 (two-kernel decode: a gather stage appears); the product build must not"""
    monkeypatch.setenv("DABPHY_FUSED_MSC", "0")
    monkeypatch.setenv("DABPHY_SP_MAX_CW", "0")
    P.check_mixed_ensemble(lambda **kw: capi.DabPhy(lib_path=LIB, decode_shape=1, **kw), F=4, nf=11, expect_fused=True)


@pytest.mark.parametrize("snr,cfo,delay,nf,lockstep", [(13, 137, 1000, 8, True), (20, -400, 333, 8, False)])
def test_stream(emu_product, snr, cfo, delay, nf, lockstep):
    P.check_stream_vs_oracle(factory, snr, cfo, delay, nf, lockstep)


def test_benchmark_handle_configuration_small(emu_product):
    from welle_io_amd import workload
    P.check_bench_config(capi, LIB, 3, 3, 1, check_ens=[0, 2], n_steps=4, demod_chunk=25, device="cpu", subs_idx=(0, 17),
                         base=workload.make_base_streams(2, workload.REC_FRAMES, seed0=0), expect_chunk=25, decode_shape=1)


def test_default_choice_of_the_viterbi_kernel(emu_product, monkeypatch):
    """dabphy_config.decode_shape = 0 on the product build (no environment override exists in it): one ensemble, one frame per call --
    the live receiver's shape -- is decoded state-parallel, one code word per wavefront (76 code words), and so are the per-frame seams;
    above 1 024 code words per call two code words share a wavefront and the traceback is a pass of its own; a batch beyond 40 960 code
    words takes the lane-per-code-word kernel.  Bytes against the oracle where checked."""
    monkeypatch.delenv("DABPHY_SP_MAX_CW", raising=False)
    from welle_io_amd import synth
    import refapi as R
    import numpy as np
    x, tx = synth.make_stream(4, snr_db=15, cfo_hz=30, delay=77, return_tx=True, seed=9)
    subs = [tx.subchs[3], tx.subchs[10]]
    o = R.orc_receiver_run(x, subchs=subs)
    d = factory(n_ensembles=1, max_frames=1, want_constellation=False)
    try:
        assert d.last_decode_plan() == (0, 0)
        d.stream_upload(x)
        d.set_subchannels([(s.subch_id, s.start_cu, s.size_cu, P.dev_prot(d, s)) for s in subs])
        fibs = []
        for _ in range(3):
            d.process(1)
            assert d.last_decode_plan() == (3, 1), d.last_decode_plan()        # k_viterbi_sp, the one MSC class in the launch
            if d.frame_info()[0, 0]["valid"] == 1:
                fibs.append(d.fibs()[0][0, 0])
        assert len(fibs) >= 2 and np.array_equal(np.array(fibs), o["fib"][:12 * len(fibs)].reshape(len(fibs), 12, 33)[:, :, 1:])
        P.check_viterbi(d, 768, 3, seed=5, kind="uniform")                      # a seam call of three code words: one wavefront each
        P.check_fic_arbitrary_int8(d, n_frames=1)
    finally:
        d.close()
    got = []
    for shape in (0, 1):                                                        # 4 x 4 frames x (4 FIC + 72 MSC) = 1 216 code words: k_viterbi_sp2 + k_traceback_sp2; the lane-per-code-word kernel beside it
        mid = factory(n_ensembles=4, max_frames=4, want_constellation=False, decode_shape=shape)
        try:
            mid.stream_upload(np.tile(x, (4, 1)))
            mid.set_subchannels([(s.subch_id, s.start_cu, s.size_cu, P.dev_prot(mid, s)) for s in tx.subchs])
            mid.process(4)
            assert mid.last_decode_plan() == ((2, 1) if shape == 0 else (1, 1)), mid.last_decode_plan()
            got.append([mid.fibs()[0].copy()] + [mid.msc(i)[0][:, :8].copy() for i in (0, 7, 17)])     # the rows the batch decoded
            assert (mid.msc_rows == 8).all()
        finally:
            mid.close()
    assert all(np.array_equal(a, b) for a, b in zip(*got)) and got[0][0].any() and got[0][3].any()
    big = factory(n_ensembles=135, max_frames=4, want_constellation=False)      # 135 x 4 frames x (4 FIC + 72 MSC) = 41 040 code words with 18 sub-channels
    try:
        big.stream_upload(np.tile(x, (135, 1)))
        big.set_subchannels([(s.subch_id, s.start_cu, s.size_cu, P.dev_prot(big, s)) for s in tx.subchs])
        big.process(4)
        assert big.last_decode_plan() == (1, 1), big.last_decode_plan()
    finally:
        big.close()
