"""The benchmark's looping recordings are at least one batch long (welle_io_amd/workload.py: rec_frames_for): a step must read
every sample once -- a ring shorter than the batch lets the device's Infinity Cache serve the second reads and flatters the FFT stage
(DESIGN.md section 6)."""
import os
import sys

import pytest

from conftest import EMU_LIB, ROOT
from welle_io_amd import capi, workload


def test_recording_is_at_least_one_batch_and_whole_periods():
    for f in (1, 16, 20, 21, 32, 40, 41, 64):
        n = workload.rec_frames_for(f)
        assert n >= f and n % 20 == 0 and n - f < 20      # whole superframes (5 frames) and interleaver periods (4 frames), no more than needed


@pytest.mark.parametrize("kind", ["drift", "low_snr"])
def test_channel_legs_of_the_bench_on_the_execution_model(emu, kind):
    """bench.py's `extras.drift` / `extras.low_snr` legs (drifting sample clocks: non-looping resampled streams; 6-10 dB) at a size the
    kernels' CPU execution model finishes: the leg runs, its own parity leg against the oracle is green, and under drift the serial
    synchroniser chain really took over from the wide pass"""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    out = bench.channel_leg(capi, workload, torch, EMU_LIB, 2, 2, 2, 0, 1, kind, device="cpu")
    assert out.get("parity") is True, out
    assert out["frames_per_step"] == 4, out
    if kind == "drift":
        assert out["wide_sync_stats"]["passes"] >= 2 and out["per_ensemble"]["ppm"][0] >= 1.0, out
