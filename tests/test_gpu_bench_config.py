"""Parity of the configuration bench.py times (welle_io_amd/workload.py builds signal and handle for both): B = 256 ensembles x F = 32
frames per step (and F = 20, the depth of round 1 and most of round 2), looping HBM-resident ring, per-ensemble carrier offset and noise, coarse corrector enabled, pipelined synchroniser
(all schedules), 3-chunk demod grid (demod_chunk = 25, chosen by dabphy_create for B x F >= 1024), all 18 sub-channels, superframe
filter inside process().  Checked against the oracle on the very same samples for ensembles spread over the batch -- first, last, ones
whose 64-codeword Viterbi groups straddle an ensemble boundary (an ensemble is 1440 codewords = 22.5 groups: every odd one does):
FIB bytes + CRC flags, correctors, MSC bytes of three sub-channels, all 230 400 soft bits of one frame per step, superframe totals."""
import os

import pytest

import parity_cases as P
from conftest import GPU_LIB
from welle_io_amd import capi, workload

pytestmark = pytest.mark.gpu

_base = {}


def base_streams(frames_per_step=20):
    """the looping recordings bench.py builds for this batch depth (at least one batch long: workload.rec_frames_for)"""
    n = workload.rec_frames_for(frames_per_step)
    if n not in _base:
        _base[n] = workload.make_base_streams(4, n, seed0=0)
    return _base[n]


@pytest.mark.parametrize("mode", [1])             # (schedules 2 and 3: the 20-frame twin below and tests/test_emu_stream.py)
def test_benchmarked_configuration(gpu, mode):
    P.check_bench_config(capi, GPU_LIB, 256, 32, mode, check_ens=[0, 1, 77, 128, 129, 255], n_steps=3, base=base_streams(32), expect_chunk=25)


@pytest.mark.parametrize("mode", [2, 3])
def test_benchmarked_configuration_20_frames(gpu, mode):
    """the batch depth of round 1 and most of round 2 (--frames 20: 5.6 Viterbi groups per SIMD)"""
    P.check_bench_config(capi, GPU_LIB, 256, 20, mode, check_ens=[0, 3, 129, 131, 255], n_steps=3, base=base_streams(), expect_chunk=25)


def test_benchmarked_configuration_one_work_group_per_frame(gpu):
    """demod_chunk = 75 (one work-group walks all symbols of a frame) forced onto the big batch"""
    P.check_bench_config(capi, GPU_LIB, 256, 32, 1, check_ens=[0, 129, 255], n_steps=2, base=base_streams(32), expect_chunk=75, demod_chunk=75)


def test_batch_beyond_18641_frame_slots(gpu):
    """520 ensembles x 32 frames: 520 x (32 + 6) frame slots of soft bits = 4.55 GB.  Round 3's fused kernel addressed that ring with
    32-bit offsets from its start and decoded every ensemble from 504 on from the rows of the first ones (18 641 x 230 400 = 2^32); the
    offsets now start at the ring slice of the wave's first ensemble.  Ensembles on both sides of the old limit and the last one, the
    whole check of the benchmarked configuration (FIBs, correctors, soft bits, MSC bytes, superframe totals of all 18 sub-channels)"""
    P.check_bench_config(capi, GPU_LIB, 520, 32, 1, check_ens=[0, 503, 504, 519], n_steps=2, base=base_streams(32), expect_chunk=25)


def hetero_base(frames_per_step):
    lib = capi.load_library(GPU_LIB)
    n = workload.rec_frames_for(frames_per_step)
    key = ("hetero", n)
    if key not in _base:
        _base[key] = workload.make_base_streams(2, n, seed0=50, subchs=workload.hetero_subchannels(lib))
    return _base[key]


def test_heterogeneous_multiplex_at_the_benchmarked_geometry(gpu):
    """bench.py's `hetero` leg: 256 ensembles x 32 frames of a multiplex as they are on air (15 sub-channels, 6 protection classes incl.
    EEP-B and UEP, code words of 192 .. 3072 bits): all classes and the FIC in ONE fused launch, code word groups that straddle
    sub-channels and ensembles in several classes at once.  Every sub-channel's bytes of the first, two middle and the last ensemble
    (and FIBs, correctors, soft bits, superframe totals) against the oracle"""
    P.check_bench_config(capi, GPU_LIB, 256, 32, 1, check_ens=[0, 127, 128, 255], n_steps=2, base=hetero_base(32), expect_chunk=25, subs_idx=tuple(range(15)))


def test_independent_ensembles_at_the_benchmarked_geometry(gpu):
    """bench.py's `mixed_layouts` leg: 256 ensembles x 32 frames, ensemble b on multiplex b % 5 (canonical, heterogeneous, two random
    ones, canonical) with ITS OWN selection (all / all / all / every other service / none) -- what a batch of real receivers is
    (msc-handler.cpp:61-127).  20 protection classes whose pair tables skip ensembles, code word groups that straddle pairs of
    different ensembles; every selected sub-channel's bytes, FIBs, correctors and superframe totals of ten ensembles spread over the
    batch (each layout at least once, first and last) against the oracle"""
    P.check_mixed_layouts(capi, GPU_LIB, 256, 32, check_ens=[0, 1, 2, 3, 4, 128, 255], n_steps=2)


def test_demod_chunk_sizes(gpu):
    P.check_demod_chunks(lambda **kw: capi.DabPhy(lib_path=GPU_LIB, **kw))
