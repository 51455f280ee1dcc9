"""The benchmark's looping recordings are at least one batch long (welle_io_amd/workload.py: rec_frames_for): a step must read
every sample once -- a ring shorter than the batch lets the device's Infinity Cache serve the second reads and flatters the FFT stage
(DESIGN.md section 6)."""
from welle_io_amd import workload


def test_recording_is_at_least_one_batch_and_whole_periods():
    for f in (1, 16, 20, 21, 32, 40, 41, 64):
        n = workload.rec_frames_for(f)
        assert n >= f and n % 20 == 0 and n - f < 20      # whole superframes (5 frames) and interleaver periods (4 frames), no more than needed
