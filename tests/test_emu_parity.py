"""CPU suite: the kernel SOURCES, compiled unchanged for the tests/hipemu execution model, against the oracle.
Checks indexing / barrier / packing logic without a GPU; the GPU suite repeats these on the device."""
import numpy as np
import pytest

import parity_cases as P


@pytest.mark.parametrize("nbits,kind", [(768, "uniform"), (1536, "coded"), (192, "extreme"), (768, "zeros"), (2304, "coded")])
def test_viterbi(emu, nbits, kind):
    P.check_viterbi(emu, nbits, 70 if nbits < 2000 else 5, seed=nbits, kind=kind)


@pytest.mark.parametrize("args", [("eep", 64, 0, 3), ("eep", 8, 0, 2), ("eep", 32, 1, 1), ("eep", 128, 0, 4), ("uep", 80, 1, 0), ("uep", 32, 5, 0)])
def test_msc_deconvolve(emu, args):
    P.check_msc_deconvolve(emu, *args, n=3, seed=7)


def test_fic(emu):
    assert P.check_fic(emu, 3, snr_db=9, seed=4) > 0


def test_fic_arbitrary_int8(emu):
    P.check_fic_arbitrary_int8(emu)


def test_demod(emu):
    P.check_demod(emu, 2, snr_db=14, seed=2)


def test_demod_degenerate_magnitudes(emu):
    P.check_demod_degenerate(emu)


def test_demod_zero_carriers(emu):
    """r1 == 0 (the reference's inf*0 -> NaN -> int8 case, SURVEY C-3): all-zero symbols give soft bit 0"""
    frames = np.zeros((1, 2048 + 75 * 2552), np.complex64)
    soft, con, _ = emu.demod_frames(frames)
    assert not soft.any()


def test_seams_state_parallel(emu):
    """the seams of INTEGRATION.md level 2 (Viterbi::deconvolve, Protection::deconvolve, FicHandler::processFicBlock) decode small calls
    with one wavefront per code word (k_viterbi_sp, kinds 1 and 2; dabphy_config.decode_shape = 2 here, the default's choice on the device):
    arbitrary int8 input incl. -128 (the clamp of viterbi.cpp:233-236), every EEP / UEP profile kind, the FIC"""
    from welle_io_amd import capi
    import conftest
    d = capi.DabPhy(lib_path=conftest.EMU_LIB, decode_shape=3)              # round 4's kernel (one code word per wavefront) stays selectable
    try:
        P.check_viterbi(d, 192, 7, seed=6, kind="extreme")
        P.check_msc_deconvolve(d, "uep", 80, 1, 0, 3, seed=11)
        P.check_fic_arbitrary_int8(d, n_frames=1)
    finally:
        d.close()
    d = capi.DabPhy(lib_path=conftest.EMU_LIB, decode_shape=2)              # k_viterbi_sp2: two code words per wavefront, odd counts included
    try:
        P.check_viterbi(d, 768, 9, seed=5, kind="uniform")
        P.check_viterbi(d, 192, 70, seed=6, kind="extreme")
        P.check_viterbi(d, 2304, 3, seed=7, kind="coded")
        P.check_viterbi(d, 32, 5, seed=8, kind="uniform")                 # 38 trellis steps: not a multiple of six -> the lane-per-code-word kernel
        P.check_msc_deconvolve(d, "eep", 64, 0, 3, 5, seed=9)
        P.check_msc_deconvolve(d, "eep", 32, 1, 1, 4, seed=10)
        P.check_msc_deconvolve(d, "uep", 80, 1, 0, 3, seed=11)
        assert P.check_fic(d, 3, 14, seed=12) > 0
        P.check_fic_arbitrary_int8(d)                                     # -128 through kind 1 (round 4 clamped only kind 2: ADVICE)
    finally:
        d.close()


def test_error_behaviour(emu):
    from welle_io_amd import capi
    import conftest
    P.check_error_behaviour(lambda **kw: capi.DabPhy(lib_path=conftest.EMU_LIB, **kw))


def test_timing_driver_refuses_a_stale_launch(emu):
    from welle_io_amd import capi
    import conftest
    P.check_timing_driver_refuses_a_stale_launch(lambda **kw: capi.DabPhy(lib_path=conftest.EMU_LIB, **kw))


def test_demod_chunk_sizes(emu):
    from welle_io_amd import capi
    import conftest
    P.check_demod_chunks(lambda **kw: capi.DabPhy(lib_path=conftest.EMU_LIB, **kw))


def test_reed_solomon_random_error_patterns(emu):
    P.check_rs_random(emu, n_sf=120)
