"""CPU suite: multipath, sampling-clock drift and fading through the kernel SOURCES (tests/hipemu execution model) against the oracle.
The reference's own soak harness exercises exactly these (welle-cli/tests.cpp:305-370, phasereference.cpp:73-256); here they also
break the wide synchroniser pass's prediction in most batches, so the hand-over to the frame-by-frame chain and the exact-batch replay
run under load.  tests/test_gpu_impairments.py repeats a larger matrix on the device."""
import pytest

import parity_cases as P
from conftest import EMU_LIB
from welle_io_amd import capi


def factory(**kw):
    return capi.DabPhy(lib_path=EMU_LIB, **kw)


# (channel, frames per call, pipeline schedule, FFT placement method): every channel, every schedule, every placement method once
CASES = [("ppm+60", 4, 1, 2), ("ppm-100", 3, 0, 1), ("echo300", 5, 2, 0), ("pre-echo", 4, 3, 1), ("sfn3", 6, 1, 0), ("echo600", 3, 2, 1),
         ("fade7", 5, 0, 2), ("ppm+fade", 4, 1, 0)]


@pytest.mark.parametrize("channel,F,schedule,placement", CASES)
def test_impaired_stream(emu, channel, F, schedule, placement):
    P.check_impaired_stream(factory, channel, F, schedule, placement)


def test_pre_echo_defeats_threshold_placement(emu):
    """ThresholdBeforePeak on a channel whose strongest path is not the first: the reference loses the FIC entirely (window indices 0,
    1008, ...); so must we, attempt for attempt -- and the exact-batch replay runs in every batch"""
    P.check_impaired_stream(factory, "pre-echo", 4, 1, 2, nf=14)


def test_small_batch_with_drifting_ensembles(emu):
    """the bench handle (pipelined, superframe filter inside process) over ensembles whose sampling clocks drift apart"""
    # (two ensembles here, the other two channels of the device twin in the next test: the execution model and the oracle run on one core)
    P.check_bench_config(capi, EMU_LIB, 2, 4, 1, check_ens=[0, 1], n_steps=3, demod_chunk=25, device="cpu", subs_idx=(0, 7, 17),
                         channels=[dict(ppm=60.0), dict(ppm=40.0, fade=(0.3, 7.0), echoes=[(150, 0.5j)])],
                         min_wide_fallbacks=2)


def test_find_chain_follows_drifting_windows(emu):
    """... with carrier offsets small enough for the fine corrector to stand still (it steps by (int16)(0.1 x residual Hz),
    ofdm-processor.cpp:450-451): the window searches of a batch then run in the FIND CHAIN (k_sync_find_chain: one after the other, each
    from the position the previous one really found; the cyclic-prefix sums of all of them at once), not frame by frame through the
    serial chain -- which still takes the acquisition and the frames in which a corrector moved"""
    P.check_bench_config(capi, EMU_LIB, 2, 4, 1, check_ens=[0, 1], n_steps=3, demod_chunk=25, device="cpu", subs_idx=(0, 7, 17),
                         channels=[dict(ppm=-100.0), dict(ppm=-30.0)],
                         min_wide_fallbacks=1, min_chain_frames=10, cfo_max_hz=4.0)
