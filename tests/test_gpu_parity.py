"""GPU suite (-m gpu): the product library libdabphy_hip.so on a real MI355X, through the C ABI, against the
oracle on the same seeded inputs (bit-exact), plus size-independent properties at BASELINE batch sizes."""
import numpy as np
import pytest

import parity_cases as P
import refapi as R
from welle_io_amd import synth

pytestmark = pytest.mark.gpu


def test_device_is_gfx950(gpu):
    assert "gfx950" in gpu.device_name


def test_libm_restatements_on_device(gpu):
    """hypotf / atan2f / divide as the device computes them == host libm (via the SNR + soft-bit paths below);
    here: the magnitude path in isolation through a frame whose soft bits exercise 127/x rounding"""
    P.check_demod(gpu, 1, snr_db=3, seed=11)


@pytest.mark.parametrize("nbits,kind,n", [(768, "uniform", 300), (768, "coded", 130), (1536, "coded", 200), (1536, "extreme", 64),
                                          (192, "uniform", 70), (9216, "coded", 3), (768, "zeros", 65)])
def test_viterbi(gpu, nbits, kind, n):
    P.check_viterbi(gpu, nbits, n, seed=nbits + n, kind=kind)


@pytest.mark.parametrize("args", [("eep", 64, 0, 3), ("eep", 8, 0, 2), ("eep", 32, 1, 1), ("eep", 128, 0, 4), ("eep", 192, 0, 1),
                                  ("uep", 80, 1, 0), ("uep", 32, 5, 0), ("uep", 384, 1, 0)])
def test_msc_deconvolve(gpu, args):
    P.check_msc_deconvolve(gpu, *args, n=67, seed=7)


@pytest.mark.parametrize("snr_db", [None, 20, 9, 5])
def test_fic(gpu, snr_db):
    n_ok = P.check_fic(gpu, 6, snr_db=snr_db, seed=4)
    if snr_db is None or snr_db >= 9:
        assert n_ok == 72


@pytest.mark.parametrize("snr_db,early", [(None, 100), (25, 100), (13, 199), (6, 0), (13, 400)])
def test_demod_soft_bits_bit_exact(gpu, snr_db, early):
    P.check_demod(gpu, 3, snr_db=snr_db, seed=2, early=early)


def test_demod_snr_report(gpu):
    """12 frames: OfdmDecoder reports SNR on the 11th (ofdm-decoder.cpp:155-158)"""
    soft = P.check_demod(gpu, 12, snr_db=17, seed=8)
    assert soft.shape[0] == 12


def test_demod_degenerate_magnitudes(gpu):
    P.check_demod_degenerate(gpu)


def test_div127_exhaustive(gpu):
    """the reciprocal-based 127/x of the demapper equals the IEEE quotient for every float in [2^-100, 2^100]"""
    bad4, bad6, tried = gpu.selftest_div127()
    assert tried == 200 * 2 ** 23 + 1
    assert bad6 == 0, "6-instruction variant: %d mismatches" % bad6
    assert bad4 == 0, "4-instruction variant (the one k_demod is built with): %d mismatches" % bad4


def test_unit_twiddle_product_exhaustive(gpu):
    """the fused product by tw[0] = (1, +-0) in the demod kernel's first two FFT passes returns the bits of the two multiplications and
    the addition it replaces: zero signs, denormals, infinities, NaNs"""
    bad, tried = gpu.selftest_unit_twiddle()
    assert tried == 4 * 2 ** 32
    assert bad == 0, "%d of %d products differ" % (bad, tried)


def test_demod_zero_carriers(gpu):
    frames = np.zeros((1, 2048 + 75 * 2552), np.complex64)
    soft, con, _ = gpu.demod_frames(frames)
    assert not soft.any()


def test_demod_then_fic_roundtrip_full_batch(gpu):
    """size-independent property at batch scale: encode -> modulate -> demod -> FIC decode returns the transmitted FIBs
    for every frame of a 64-frame batch (clean channel), CRC flags all set"""
    nf = 64
    x, tx = synth.make_stream(nf + 1, snr_db=None, seed=21, return_tx=True)
    frames = P.cut_frames(x, nf)
    soft, _, _ = gpu.demod_frames(frames, want_con=False)
    fib, ok, ratio = gpu.fic_decode(soft[:, :3].reshape(nf, 9216))
    assert ok.all() and ratio == 100
    sent = np.frombuffer(b"".join(b"".join(f) for f in tx.fib_log[:nf]), np.uint8).reshape(nf, 12, 32)
    assert np.array_equal(fib, sent)


@pytest.mark.parametrize("shape", [1, 2, 3])
def test_seams_with_either_decoder(gpu, shape):
    """the seams of INTEGRATION.md level 2 with both Viterbi kernels forced (the default decodes these small calls state-parallel: every
    other seam test of this file): arbitrary int8 input incl. -128, EEP / UEP profiles, the FIC"""
    from welle_io_amd import capi
    import conftest
    d = capi.DabPhy(lib_path=conftest.GPU_LIB, decode_shape=shape)
    try:
        P.check_viterbi(d, 768, 90, seed=5, kind="uniform")
        P.check_viterbi(d, 192, 700, seed=6, kind="extreme")
        P.check_viterbi(d, 9216, 5, seed=7, kind="coded")
        P.check_viterbi(d, 32, 5, seed=8, kind="uniform")
        P.check_msc_deconvolve(d, "eep", 64, 0, 3, 40, seed=9)
        P.check_msc_deconvolve(d, "uep", 80, 1, 0, 30, seed=11)
        assert P.check_fic(d, 5, 14, seed=12) > 0
        P.check_fic_arbitrary_int8(d, n_frames=7)                         # -128 through the FIC seam, whichever kernel takes it
    finally:
        d.close()


def test_error_behaviour(gpu):
    from welle_io_amd import capi
    import conftest
    P.check_error_behaviour(lambda **kw: capi.DabPhy(lib_path=conftest.GPU_LIB, **kw))


def test_timing_driver_refuses_a_stale_launch(gpu):
    from welle_io_amd import capi
    import conftest
    P.check_timing_driver_refuses_a_stale_launch(lambda **kw: capi.DabPhy(lib_path=conftest.GPU_LIB, **kw))


def test_reed_solomon_random_error_patterns(gpu):
    P.check_rs_random(gpu, n_sf=600)
