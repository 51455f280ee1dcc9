"""GPU suite: parity under the channel impairments the reference's soak harness is built for (welle-cli/tests.cpp:305-370:
multipath, comparison of the FFT placement methods, phasereference.cpp:73-256) plus sampling-clock drift and fading -- every
channel x {3, 4, 5, 6} frames per call x pipeline schedules 0-3 x all three placement methods in a covering design, and one batch of
the benchmark's geometry (256 x 32) whose ensembles drift apart so that the wide synchroniser pass falls back in every batch.
Everything is compared with the oracle with NO tolerance (FIBs, CRC flags, correctors, null symbols, every soft bit, MSC bytes)."""
import pytest

import parity_cases as P
from conftest import GPU_LIB
from welle_io_amd import capi

pytestmark = pytest.mark.gpu


def factory(**kw):
    return capi.DabPhy(lib_path=GPU_LIB, **kw)


CHANNELS = list(P.CHANNELS)
FS = (3, 4, 5, 6)
# channel c with placement method j: frames per call and schedule rotate so that every value meets every channel and method
CASES = [(ch, FS[(c + j) % 4], (c + 2 * j + 1) % 4, j) for c, ch in enumerate(CHANNELS) for j in range(3)]


@pytest.mark.parametrize("channel,F,schedule,placement", CASES)
def test_impaired_stream(gpu, channel, F, schedule, placement):
    P.check_impaired_stream(factory, channel, F, schedule, placement)


@pytest.mark.parametrize("channel,schedule", [("ppm-100", 1), ("pre-echo", 2), ("ppm+fade", 3)])
def test_impaired_stream_one_frame_per_call(gpu, channel, schedule):
    """the real-time facade's mode (one frame per call: no wide pass, no replay) on the same channels"""
    P.check_impaired_stream(factory, channel, 1, 0, 2, lockstep=True, nf=10)


@pytest.mark.parametrize("channel,placement", [("echo300", 1), ("ppm+60", 2), ("sfn3", 0)])
def test_impaired_stream_low_snr(gpu, channel, placement):
    """... at 7 dB, where the FIC ratio hovers around the coarse corrector's 50 % line: the exact-batch replay under multipath / drift"""
    P.check_impaired_stream(factory, channel, 4, 1, placement, snr_db=7, nf=22)


def test_find_chain_follows_drifting_windows_at_the_benchmark_geometry(gpu):
    """256 ensembles x 32 frames per call, the bench handle: recordings through +60 / -100 / +40 ppm (with fading and an echo) / -30 ppm,
    so the window index moves in every frame, with carrier offsets small enough for the fine correctors to stand still (correctors that
    are still converging under drift: test_impaired_stream's ppm cases): the window searches of every
    batch run in the find chain (k_sync_find_chain: each from the position the previous one really found; all cyclic-prefix sums at once);
    every FIB, corrector, soft bit, MSC byte and superframe total of the checked ensembles equals the oracle's"""
    P.check_bench_config(capi, GPU_LIB, 256, 32, 1, check_ens=[0, 1, 2, 255], n_steps=2,
                         channels=[dict(ppm=60.0), dict(ppm=-100.0), dict(ppm=40.0, fade=(0.3, 7.0), echoes=[(150, 0.5j)]), dict(ppm=-30.0)],
                         min_wide_fallbacks=1, min_chain_frames=256 * 28, cfo_max_hz=4.0)
