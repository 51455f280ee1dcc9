"""GPU suite: streaming receiver on the device vs the oracle (bit-exact FIBs, CRC flags, MSC bytes, soft bits,
constellation taps, correctors), batch mode and lock-step mode, single and multiple ensembles."""
import numpy as np
import pytest

import parity_cases as P
from conftest import GPU_LIB
from welle_io_amd import capi, synth

pytestmark = pytest.mark.gpu


def factory(**kw):
    return capi.DabPhy(lib_path=GPU_LIB, **kw)


def factory_lane_per_codeword(**kw):
    """dabphy_config.decode_shape = 1: the throughput kernel (k_viterbi_fused) also for batches the default would decode state-parallel"""
    return capi.DabPhy(lib_path=GPU_LIB, decode_shape=1, **kw)


def factory_state_parallel(**kw):
    """dabphy_config.decode_shape = 2: the state-parallel kernel (k_viterbi_sp2: two code words per wavefront) whatever the batch size"""
    return capi.DabPhy(lib_path=GPU_LIB, decode_shape=2, **kw)


def factory_state_parallel_r4(**kw):
    """dabphy_config.decode_shape = 3: round 4's state-parallel kernel (k_viterbi_sp: one code word per wavefront)"""
    return capi.DabPhy(lib_path=GPU_LIB, decode_shape=3, **kw)


@pytest.mark.parametrize("snr,cfo,delay,nf,lockstep", [(25, 0, 0, 22, False), (None, 0, 0, 9, False), (13, 137, 1000, 14, True), (20, 2300, 0, 12, True),
                                                     (20, -400, 333, 10, True), (None, 17400, 0, 8, True), (10, -1000, 0, 8, True), (8, 60, 77, 12, False)])
def test_stream(gpu, snr, cfo, delay, nf, lockstep):
    P.check_stream_vs_oracle(factory, snr, cfo, delay, nf, lockstep)


def test_ensembles_are_independent(gpu):
    """16 copies of one stream decoded side by side give 16 identical, oracle-exact results"""
    P.check_stream_vs_oracle(factory, 15, 40, 123, 10, False, B=16, F=4)


def test_full_ensemble_roundtrip(gpu):
    """all 18 sub-channels of the canonical ensemble, clean channel: every decoded byte equals the transmitted payload"""
    nf = 12
    x, tx = synth.make_stream(nf, snr_db=None, return_tx=True, seed=9)
    logs = P.run_stream(factory, x, tx.subchs, 5, 10)
    L = logs[0]
    assert len(L["fib"]) == 10 and np.array(L["ok"]).all()
    for i, s in enumerate(tx.subchs):
        got = b"".join(L["msc"][i])
        pay = b"".join(tx.payload_log[s.subch_id])
        assert len(got) >= 24 * s.frame_bytes and got in pay


def test_big_batch_tiled_gather(gpu):
    P.check_stream_vs_oracle(factory, 14, 30, 200, 43, False, F=20)


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_pipelined_sync(gpu, mode):
    """batch k+1 is synchronised on a second stream while batch k is decoded (queued behind the demod kernel, or at once): same
    bytes as the serial order"""
    P.check_stream_vs_oracle(factory, 16, -20, 50, 26, False, F=4, pipeline_sync=mode, disable_coarse=True, B=3)


def test_stream_without_constellation(gpu):
    """the demod kernel variant without the constellation tap (what the throughput path runs)"""
    P.check_stream_vs_oracle(factory, 12, 75, 10, 10, False, con=False)


@pytest.mark.parametrize("fmt", ["u8", "s8", "s16le", "s16be"])
def test_live_ring_raw_formats(gpu, fmt):
    """samples appended to the library's ring in the reference's file formats, converted on the device"""
    P.check_live_raw_vs_oracle(factory, fmt)


def test_superframe_filter(gpu):
    """DAB+ superframe synchronisation, Reed-Solomon and AU CRCs on the device = SuperframeFilter::Feed, state carried over batches"""
    got = P.check_superframes_vs_oracle(factory, nf=20)
    ev = got[0][0]
    assert any(e[0] > 0 for e in ev) and any(e[1] and e[2] and e[6] != 7 for e in ev) and any(not e[2] for e in ev[2:])   # corrections, a broken AU, a lost sync


@pytest.mark.parametrize("F,damage_q", [(2, (1, 7)), (3, (3, 9)), (5, (4, 12))])
def test_superframe_filter_where_the_damage_falls(gpu, F, damage_q):
    """a broken and a lost superframe at other places and other batch depths (F = 2 with superframe 7 lost: the wide pass starts from a full
    window that had failed and is accepted): events, corrected superframes and totals still the oracle's"""
    P.check_superframes_vs_oracle(factory, F=F, nf=22, B=1, damage_q=damage_q, auto_modes=(True,))


def test_superframe_filter_other_bit_rates(gpu):
    """the filter's other instances: 128 / 192 / 256 kbit/s (superframes of 1920 / 2880 / 3840 bytes: 16 - 32 code words, rows of more than
    one LDS-DMA request) and 32 kbit/s (4 code words: half a syndrome round), damaged superframes included, both ways through a batch"""
    from welle_io_amd import synth
    ens = [synth.SubchannelCfg(1, 0, 128), synth.SubchannelCfg(2, 96, 192), synth.SubchannelCfg(3, 240, 256), synth.SubchannelCfg(4, 432, 32)]
    P.check_superframes_vs_oracle(factory, nf=20, B=1, ensemble=ens, pick=(0, 1, 2, 3), auto_modes=(True,))


@pytest.mark.parametrize("F,nf", [(4, 11), (1, 7), (2, 9), (3, 10), (5, 14), (8, 19), (15, 33)])
def test_mixed_protection_classes(gpu, F, nf):
    """1 .. 15 frames per call = 4 .. 60 CIFs per sub-channel: a wave's 64 code words span up to 17 (ensemble, sub-channel) pairs -- the
    144- and 324-row builds of the fused kernel; no separate gather stage at any batch depth"""
    P.check_mixed_ensemble(factory_lane_per_codeword, F=F, nf=nf, expect_fused=True)


@pytest.mark.parametrize("F,nf", [(4, 11), (16, 36), (1, 7), (7, 17)])
def test_mixed_protection_classes_state_parallel(gpu, F, nf):
    """the same ensemble (EEP A/B, UEP, 8 .. 384 kbit/s: code words of 192 .. 9216 bits, all three LDS sizes of the kernel) through
    k_viterbi_sp2 + k_traceback_sp2: two code words per wavefront, lanes = trellis states, decisions as history rows, the traceback a pass
    of its own (lane = code word, wave = a stretch of the code word, guessed entry states checked).  (Every other stream test of this file
    runs small batches too, hence the state-parallel kernels: the default picks them below 40 960 code words per call.)"""
    P.check_mixed_ensemble(factory_state_parallel, F=F, nf=nf, expect_fused=True)


@pytest.mark.parametrize("shape", [1, 2])
def test_stream_with_either_decoder(gpu, shape):
    """the canonical ensemble through both Viterbi kernels explicitly (soft bits, FIBs, MSC bytes of all 18 sub-channels vs the oracle)"""
    f = factory_lane_per_codeword if shape == 1 else factory_state_parallel
    P.check_stream_vs_oracle(f, 11, 80, 250, 9, False, F=3)


def test_lane_exchanges_of_the_state_parallel_kernel(gpu):
    """v_permlane32_swap / v_permlane16_swap / bank-masked row DPP / quad_perm DPP / v_readlane as k_viterbi_sp uses them, against plain
    shuffles on the device (the GPU-less execution model stands in for exactly these forms)"""
    bad, n = gpu.selftest_pair_exchange()
    # (13 checks per lane and round of k_viterbi_sp's forms, 8 of k_viterbi_sp2's swap16 / swap32 / partner<3..0>)
    assert bad == 0 and n == 8 * 64 * 16 * (13 + 8), (bad, n)


def test_shallow_batch_above_the_state_parallel_limit(gpu):
    """288 ensembles x 4 frames of the mixed ensemble = 46 080 code words per call: too many for the state-parallel kernels' default
    limit (40 960), too shallow for the 96-row build: the 144-row build of the fused kernel as the DEFAULT choice"""
    P.check_mixed_ensemble(factory, F=4, nf=11, B=288, expect_fused=True, check_ens=(0, 143, 287))


def test_shallow_batch_below_the_state_parallel_limit(gpu):
    """128 ensembles x 4 frames of the mixed ensemble = 20 480 code words per call, code words of 192 .. 9216 bits in nine classes: the
    DEFAULT choice is k_viterbi_sp2 with its traceback as a pass of its own"""
    P.check_mixed_ensemble(factory, F=4, nf=11, B=128, expect_fused=True, check_ens=(0, 63, 127))


def test_two_kernel_decode_beyond_the_fused_kernels_reach(gpu):
    """a handle with ring slices of 4102 frames (945 MB) per ensemble, 4 frames per call, six ensembles: the five consecutive pairs
    a wave of a one-sub-channel-per-ensemble class spans lie in five ensembles, 4.7 GB apart -- beyond the 32-bit offsets of the fused
    kernel's buffer resource -- so every class and the FIC go through k_msc_gather / k_fic_gather + k_viterbi (64-bit addresses), and
    decode the same bytes"""
    P.check_mixed_ensemble(factory_lane_per_codeword, F=4, nf=11, B=6, max_frames=4096, expect_fused=False, check_ens=(0, 3, 5))


def test_fused_decode_of_ensembles_beyond_4_gib(gpu):
    """six ensembles whose ring slices are 945 MB apart: ensemble 5's soft bits start 4.7 GB behind the ring's (round 3's fused kernel
    addressed them with 32-bit offsets from the START of the ring).  16 frames per call, two narrow sub-channels; the emulator runs
    the same case"""
    subchs = [synth.SubchannelCfg(1, 0, 32, False, 3, dabplus=False), synth.SubchannelCfg(2, 24, 8, False, 2, dabplus=False)]
    P.check_mixed_ensemble(factory_lane_per_codeword, F=16, nf=36, B=6, max_frames=4096, subchs=subchs, expect_fused=True, check_ens=(0, 4, 5))


def test_mixed_protection_classes_fused_decode(gpu):
    """16 / 20 frames per call: every class (EEP A/B, UEP, 8 .. 384 kbit/s) takes the fused kernel's 96-row build (k_viterbi_fused)"""
    P.check_mixed_ensemble(factory_lane_per_codeword, F=16, nf=36, expect_fused=True)
    P.check_mixed_ensemble(factory_lane_per_codeword, F=20, nf=45, seed=32, snr_db=9, expect_fused=True)


@pytest.mark.parametrize("method,snr,cfo", [(1, 15, 90), (1, None, -300), (0, 12, 40)])
def test_other_fft_placement_methods(gpu, method, snr, cfo):
    """RadioReceiverOptions::fftPlacementMethod: EarliestPeakWithBinning (1) and StrongestPeak (0)"""
    P.check_stream_vs_oracle(factory, snr, cfo, 211, 9, True, fft_placement=method)


@pytest.mark.parametrize("freqsync,cfo,snr", [(0, 2300, 20), (1, 2300, 20), (1, -1000, 14), (0, 17400, None)])
def test_other_freqsync_methods(gpu, freqsync, cfo, snr):
    """RadioReceiverOptions::freqsyncMethod: GetMiddle (0) and CorrelatePRS (1) drive the coarse corrector"""
    P.check_stream_vs_oracle(factory, snr, cfo, 250, 10, True, seed=50 + freqsync, freqsync=freqsync)


@pytest.mark.parametrize("ring_extra", [44500, 77777])
def test_live_ring_that_wraps_inside_the_frames(gpu, ring_extra):
    """a ring of 6 frames + 44 500 / 77 777 samples: its end falls on another OFDM symbol in every revolution -- the first symbol of a demod
    work-group's chunk (its LDS-DMA pipeline starts one symbol late) and symbols in the middle of a chunk (the pipeline is interrupted and
    picked up again); that symbol comes straight from HBM.  (Which symbols: seen by instrumenting the kernel once.)"""
    P.check_live_raw_vs_oracle(factory, "s16le", nf=20, ring_extra=ring_extra)


def test_live_ring_async_ingest(gpu):
    """dabphy_stream_write_raw_async: copy + conversion on the copy stream, dabphy_process orders itself behind them"""
    P.check_live_raw_vs_oracle(factory, "u8", asynchronous=True)


@pytest.mark.parametrize("pipelined", [False, True])
def test_tii_side_path(gpu, pipelined):
    """TIIDecoder on the device: four ensembles with different transmitter sets, sums carried across batches"""
    P.check_tii_vs_oracle(factory, pipeline_sync=pipelined)


def test_fine_corrector_interval_and_exact_paths(gpu):
    P.check_fine_corrector_paths(factory)


def test_dropout_and_relock(gpu):
    P.check_dropout_relock(factory)


def test_relock_after_long_lock(gpu):
    P.check_relock_after_long_lock(factory)


def test_lock_lost_inside_a_replayed_batch(gpu):
    P.check_lock_lost_inside_a_replayed_batch(factory)


@pytest.mark.parametrize("mode", [1, 3])
def test_lock_lost_inside_a_replayed_batch_pipelined(gpu, mode):
    P.check_lock_lost_inside_a_replayed_batch(factory, pipeline_sync=mode)


def test_fine_corrector_on_the_edge(gpu):
    P.check_fine_corrector_on_the_edge(factory)


@pytest.mark.parametrize("fmt", ["u8", "s8", "s16le", "s16be"])
def test_ingest_vs_reference_crawfile(gpu, fmt, tmp_path):
    """k_ingest pinned to the real CRAWFile::convertSamples (input/raw_file.cpp:324-366, compiled into oracle/_ref)"""
    import refapi as R
    if not R.have_ref():
        pytest.skip("oracle/_ref not built")
    P.check_ingest_vs_rawfile(factory, fmt, tmp_path)


@pytest.mark.parametrize("snr,cfo,F,seed", [(4, 300, 4, 3), (3, -1000, 4, 5), (5, 2300, 6, 7), (2, 40, 4, 9), (3, -1000, 8, 11), (4, 17400, 5, 13)])
def test_low_snr_batches_with_coarse_corrector(gpu, snr, cfo, F, seed):
    """batch mode (F > 1) with the coarse corrector enabled while the FIC decodes badly (ratio around / below 50): the corrector then
    consults the previous batch's ratio (include/dabphy.h, dabphy_process); these streams, including losses of lock inside a batch,
    either give the oracle's frames bit for bit or part from them exactly where the ratio crossed the 50 % line inside a batch"""
    P.check_stream_vs_oracle(factory, snr, cfo, 150, 25, False, F=F, seed=seed, ratio_lag_ok=True)


def test_dropout_in_batch_mode(gpu):
    """a dropout decoded four frames per call: the slot after the failed window search re-acquires at once (k_acquire is queued before
    every frame step), MSC rows stay packed, the superframe filter walks the frames that exist"""
    P.check_dropout_batch(factory)


def test_receiver_options_at_run_time(gpu):
    P.check_runtime_options(factory)


@pytest.mark.parametrize("pipeline", [False, 1, 2])
def test_wide_synchroniser_pass(gpu, pipeline):
    """all frames of a batch synchronised at once from the predicted in-lock state (k_sync_find_wide / k_sync_finish_wide /
    k_sync_validate) = the frame-by-frame chain = the oracle, bit for bit; the counters show which path produced the frames"""
    P.check_wide_sync(factory, pipeline_sync=pipeline, nf=44, F=8, B=3)


@pytest.mark.parametrize("snr,cfo,F,seed,pipeline,replay", [(3, -1000, 4, 5, False, True), (3, -1000, 4, 5, 1, True), (3, -1000, 4, 5, 2, None), (3, -1000, 4, 5, 3, None),
                                                            (4, 300, 4, 3, 1, None), (2, 40, 4, 9, 2, None), (3, -1000, 8, 11, False, None), (4, 17400, 5, 13, 1, None), (5, 2300, 6, 7, False, None)])
def test_exact_batch_mode(gpu, snr, cfo, F, seed, pipeline, replay):
    """Exact batch mode (the default): the low-SNR batch streams again, now required to equal the oracle frame for frame without any
    tolerance (a batch whose stale coarse-corrector decision can have mattered is put back and decoded a second time with the
    reference's per-frame FIC-ratio feedback); the first two are known to need that second pass"""
    P.check_exact_batch(factory, snr, cfo, F, seed, pipeline_sync=pipeline, expect_replay=replay)


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_one_frame_per_call_on_a_pipelined_schedule(gpu, mode):
    """one frame per call with the synchroniser one or two frames ahead (pipeline_sync 1-3): the second pass of exact batch mode is armed
    for it too (ofdm-processor.cpp:397-409: the coarse corrector consults the FIC ratio of the PREVIOUS frame); equal to the oracle frame
    for frame, on the stream whose coarse corrector acts on a stale ratio from frame 6 on, and on a second one at another offset"""
    logs = P.check_exact_batch(factory, 3, -1000, 1, 5, pipeline_sync=mode, nf=21)
    if mode == 3:
        assert logs[0]["replayed"] >= 1, logs[0]["replayed"]
    P.check_exact_batch(factory, 4, 17400, 1, 13, pipeline_sync=mode, nf=17)


@pytest.mark.parametrize("chunk", [1, 2])
def test_exact_batch_mode_with_short_demod_chunks(gpu, chunk):
    """the replay demodulates only the chunks that hold the FIC symbols 1 .. 3 of a frame: with one or two symbols per work-group that
    is three or two chunks (a replay that took only the first one read stale soft bits: advisor, round 2); the batch must still equal
    the oracle frame for frame and must have been decoded twice"""
    P.check_exact_batch(lambda **kw: factory(demod_chunk=chunk, **kw), 3, -1000, 4, 5, pipeline_sync=False, expect_replay=True)


def test_superframes_through_a_replayed_batch(gpu):
    """exact batch mode with the superframe filter: a 3.5 dB stream in which one batch has to be decoded a second time -- the filter's
    windows are put back with the rest of the state -- gives the oracle's superframe events, corrected superframes and totals whether
    the filter is called per sub-channel, for all of them after the batch, or rides inside dabphy_process"""
    st = {}
    P.check_superframes_vs_oracle(factory, F=3, nf=22, snr_db=3.5, seed=10, B=1, damage=True, cfo=40, stats=st)
    assert st["replayed"] >= 1 and st["replayed_auto_0"] >= 1 and st["replayed_auto_1"] >= 1, st


@pytest.mark.parametrize("snr,cfo,F,pipeline", [(4, -1000, 4, False), (3.5, -1000, 3, 2)])
def test_tii_through_replayed_batches(gpu, snr, cfo, F, pipeline):
    """exact batch mode with the TII side path: at 3.5-4 dB batches have to be decoded a second time (the TII sums are put back with the
    rest of the state); the measurements still equal the TIIDecoder restatement fed by the oracle receiver"""
    st = {}
    P.check_tii_vs_oracle(factory, F=F, nf=21, snr_db=snr, cfo=cfo, pipeline_sync=pipeline, stats=st, counts=False)
    assert st["replayed"] >= 1, st


@pytest.mark.parametrize("snr,cfo,seed,F", [(3, 300, 17, 3), (20, 37, 9, 4)])
def test_live_ring_in_batches(gpu, snr, cfo, seed, F):
    """a live ring (s16 samples written as they arrive) decoded F frames per call: slots without samples inside the batches, the wide
    synchroniser pass and its serial fall-back, at 3 dB a loss of lock with re-acquisition -- FIBs and MSC bytes = the oracle's"""
    st = {}
    P.check_live_batch_vs_oracle(factory, F=F, snr_db=snr, cfo=cfo, seed=seed, stats=st)
    assert st["wide"][0][0] >= 1, st


def test_live_ring_shorter_than_the_lock(gpu):
    """3 dB, lock lost after 12 frames in a live ring of 13: the samples sLevel would have to be replayed from are gone, the two
    bracketing replays start from 0 and from the bound of converted s16 samples (2.125, not 3e38) and meet within the history that
    is left: the re-acquisition is certified and lands where the reference's does"""
    st = {}
    P.check_live_batch_vs_oracle(factory, F=3, snr_db=3, cfo=300, seed=17, stats=st, ring_frames=13)
    assert st["lost"] >= 1 and st["relock_inexact"] == 0, st


@pytest.mark.parametrize("pipeline", [False, 1, 2, 3])
def test_exact_batch_mode_with_different_ensembles(gpu, pipeline):
    """one ensemble of three makes a batch be decoded twice: all three must still equal their own oracle runs"""
    P.check_exact_batch_mixed(factory, pipeline_sync=pipeline)


@pytest.mark.parametrize("shape", ["default", "lane", "state"])
def test_service_added_and_removed_in_mid_stream(gpu, shape):
    """dabphy_set_subchannels_ensemble between batches (MscHandler::addSubchannel / removeSubchannel on running receivers,
    msc-handler.cpp:61-127): the services that keep playing -- in the changed ensemble and in the other one -- deliver the uninterrupted
    stream's bytes and SuperframeFilter events; the added one starts like a fresh DabAudio (first frame on the 17th CIF)"""
    P.check_service_changes_in_mid_stream({"default": factory, "lane": factory_lane_per_codeword, "state": factory_state_parallel}[shape])


def test_service_changes_with_deep_batches(gpu):
    """the same with eight frames per call (the 144-row build of the fused kernel; the superframe filter's wide pass)"""
    P.check_service_changes_in_mid_stream(factory_lane_per_codeword, F=8, nf=58, add_step=2, remove_step=4)


@pytest.mark.parametrize("shape", [0, 2])          # (lane per code word: the 256 x 32 twin in test_gpu_bench_config.py)
def test_independent_ensembles_in_one_batch(gpu, shape):
    """ten ensembles, five multiplexes, five selections (all / all / all / every other service / none), four frames per call, with
    either Viterbi kernel: every selected sub-channel of every ensemble, FIBs and superframe totals against the oracle"""
    P.check_mixed_layouts(capi, GPU_LIB, 10, 4, check_ens=list(range(10)), n_steps=3, decode_shape=shape, rec_frames=20)


def test_round_4_state_parallel_kernel_is_still_there(gpu):
    """dabphy_config.decode_shape = 3 (k_viterbi_sp, one code word per wavefront): the mixed ensemble incl. its 9216-bit code words and a
    stream with services changing in mid-stream, against the oracle"""
    P.check_mixed_ensemble(factory_state_parallel_r4, F=4, nf=11, expect_fused=True)
    P.check_service_changes_in_mid_stream(factory_state_parallel_r4)


def test_medium_batch_takes_the_split_state_parallel_path(gpu):
    """16 ensembles x 8 frames x 76 code words = 9728 per call: the automatic choice is k_viterbi_sp2 with the traceback as its own
    lane-per-code-word pass (k_traceback_sp2); FIBs, correctors, soft bits and MSC bytes of every ensemble against the oracle"""
    d = capi.DabPhy(lib_path=GPU_LIB, n_ensembles=16, max_frames=8)
    d.close()
    P.check_stream_vs_oracle(factory, 15, 40, 123, 26, False, B=16, F=8, con=False)


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_service_changes_while_the_synchroniser_runs_ahead(gpu, mode):
    """selection changes between dabphy_process calls with the pipelined schedules: they apply to the batch decoded next, whatever has been
    synchronised ahead; bytes and superframe events of the services that stay = the uninterrupted oracle's"""
    P.check_service_changes_in_mid_stream(factory, pipeline_sync=mode)
