"""CPU suite: the whole streaming receiver (acquisition, PRS sync, coarse/fine tracking, demod, FIC, MSC with time
de-interleaving) compiled for tests/hipemu, against orc_receiver_run -- which tests/test_oracle_vs_ref.py pins to
the real reference end to end."""
import pytest

import parity_cases as P
from conftest import EMU_LIB
from welle_io_amd import capi


def factory(**kw):
    return capi.DabPhy(lib_path=EMU_LIB, **kw)


def factory_lane_per_codeword(**kw):
    """dabphy_config.decode_shape = 1: the throughput kernel (k_viterbi_fused) also for batches the default would decode state-parallel"""
    return capi.DabPhy(lib_path=EMU_LIB, decode_shape=1, **kw)


def factory_state_parallel(**kw):
    """dabphy_config.decode_shape = 2: the state-parallel kernel (k_viterbi_sp2: two code words per wavefront) whatever the batch size"""
    return capi.DabPhy(lib_path=EMU_LIB, decode_shape=2, **kw)


@pytest.mark.parametrize("snr,cfo,delay,nf,lockstep", [(25, 0, 0, 10, False), (13, 137, 1000, 8, True), (20, 2300, 0, 8, True), (20, -400, 333, 8, True)])
def test_stream(emu, snr, cfo, delay, nf, lockstep):
    P.check_stream_vs_oracle(factory, snr, cfo, delay, nf, lockstep)


def test_two_ensembles_in_lock_step(emu):
    P.check_stream_vs_oracle(factory, 18, 0, 500, 7, False, B=2, F=2)


def test_big_batch_tiled_gather(emu):
    """F = 20 (80 CIFs per sub-channel and batch): exercises the LDS-tiled MSC gather incl. groups straddling two sub-channels"""
    P.check_stream_vs_oracle(factory, 14, 30, 200, 43, False, F=20)


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_pipelined_sync(emu, mode):
    """throughput mode: batch k+1 synchronised ahead of batch k's decode gives the same bytes (coarse corrector off)"""
    P.check_stream_vs_oracle(factory, 16, -20, 50, 14, False, F=3, pipeline_sync=mode, disable_coarse=True)


def test_stream_without_constellation(emu):
    P.check_stream_vs_oracle(factory, 12, 75, 10, 7, False, con=False)


@pytest.mark.parametrize("fmt", ["u8", "s8", "s16le", "s16be"])
def test_live_ring_raw_formats(emu, fmt):
    """samples appended to the library's ring in the reference's file formats, converted on the device"""
    P.check_live_raw_vs_oracle(factory, fmt)


def test_superframe_filter(emu):
    """DAB+ superframe synchronisation, Reed-Solomon and AU CRCs on the device = SuperframeFilter::Feed, state carried over batches"""
    got = P.check_superframes_vs_oracle(factory, nf=20, auto_modes=(True,))     # (the GPU run checks both launch paths)
    ev = got[0][0]
    assert any(e[0] > 0 for e in ev) and any(e[1] and e[2] and e[6] != 7 for e in ev) and any(not e[2] for e in ev[2:])   # corrections, a broken AU, a lost sync


@pytest.mark.parametrize("F,damage_q", [(2, (1, 3)), (2, (1, 7)), (3, (3, 9)), (4, (2, 11)), (5, (4, 12)), (3, (5, 15)), (4, (7, 13))])
def test_superframe_filter_where_the_damage_falls(emu, F, damage_q):
    """a broken and a lost superframe at other places and other batch depths: the filter's window then re-aligns at other rows of a batch
    -- in the middle (the batch is walked frame by frame), or with its first frame (the wide pass starts from a full window that had
    failed and is accepted: F = 2 with superframe 7 lost, found by instrumenting the verdict kernel) --, events, corrected superframes and
    totals still the oracle's"""
    P.check_superframes_vs_oracle(factory, F=F, nf=22, B=1, damage_q=damage_q, auto_modes=(True,))


def test_superframe_filter_other_bit_rates(emu):
    """the filter's other instances: 128 / 192 / 256 kbit/s (superframes of 1920 / 2880 / 3840 bytes: 16 - 32 code words, rows of more than
    one LDS-DMA request) and 32 kbit/s (4 code words: half a syndrome round), damaged superframes included, both ways through a batch"""
    from welle_io_amd import synth
    ens = [synth.SubchannelCfg(1, 0, 128), synth.SubchannelCfg(2, 96, 192), synth.SubchannelCfg(3, 240, 256), synth.SubchannelCfg(4, 432, 32)]
    P.check_superframes_vs_oracle(factory, nf=20, B=1, ensemble=ens, pick=(0, 1, 2, 3), auto_modes=(True,))


@pytest.mark.parametrize("F,nf", [(4, 11), (1, 7), (3, 10)])
def test_mixed_protection_classes(emu, F, nf):
    """4 / 1 / 3 frames per call = 16 / 4 / 12 CIFs per sub-channel: a wave's 64 code words span up to 5 / 17 / 7 (ensemble,
    sub-channel) pairs -- the 144- and 324-row builds of the fused kernel; no separate gather stage at any batch depth"""
    P.check_mixed_ensemble(factory_lane_per_codeword, F=F, nf=nf, expect_fused=True)


@pytest.mark.parametrize("F,nf", [(3, 9), (1, 7)])           # (the device suite runs 4, 16 and 7 frames per call too: the execution model needs minutes for those)
def test_mixed_protection_classes_state_parallel(emu, F, nf):
    """the same ensemble (EEP A/B, UEP, 8 .. 384 kbit/s: code words of 192 .. 9216 bits, all three LDS sizes of the kernel) through
    k_viterbi_sp: one wavefront per code word, lanes = trellis states, decisions as per-lane histories, scalar traceback.  (Every other
    stream test of this file runs small batches too, hence this kernel: the default picks it below 16 384 code words per call.)"""
    P.check_mixed_ensemble(factory_state_parallel, F=F, nf=nf, expect_fused=True)


@pytest.mark.parametrize("shape", [1, 2])
def test_stream_with_either_decoder(emu, shape):
    """the canonical ensemble through both Viterbi kernels explicitly (soft bits, FIBs, MSC bytes of all 18 sub-channels vs the oracle)"""
    f = factory_lane_per_codeword if shape == 1 else factory_state_parallel
    P.check_stream_vs_oracle(f, 11, 80, 250, 9, False, F=3)


def test_two_kernel_decode_beyond_the_fused_kernels_reach(emu):
    """a handle with ring slices of 4102 frames (945 MB) per ensemble, 4 frames per call, six ensembles: the five consecutive pairs
    a wave of a one-sub-channel-per-ensemble class spans lie in five ensembles, 4.7 GB apart -- beyond the 32-bit offsets of the fused
    kernel's buffer resource -- so every class and the FIC go through k_msc_gather / k_fic_gather + k_viterbi (64-bit addresses), and
    decode the same bytes"""
    P.check_mixed_ensemble(factory_lane_per_codeword, F=4, nf=11, B=6, max_frames=4096, expect_fused=False, check_ens=(0, 3, 5))


def test_fused_decode_of_ensembles_beyond_4_gib(emu):
    """six ensembles whose ring slices are 945 MB apart: ensemble 5's soft bits start 4.7 GB behind the ring's -- round 3's fused
    kernel addressed them with 32-bit offsets from the START of the ring and decoded ensemble 5 from ensemble 0's rows.  16 frames per
    call (two segments per wave: the 96-row build), two narrow sub-channels"""
    from welle_io_amd import synth
    subchs = [synth.SubchannelCfg(1, 0, 32, False, 3, dabplus=False), synth.SubchannelCfg(2, 24, 8, False, 2, dabplus=False)]
    P.check_mixed_ensemble(factory_lane_per_codeword, F=16, nf=36, B=6, max_frames=4096, subchs=subchs, expect_fused=True, check_ens=(0, 4, 5))


@pytest.mark.parametrize("method,snr,cfo", [(1, 15, 90), (1, None, -300), (0, 12, 40)])
def test_other_fft_placement_methods(emu, method, snr, cfo):
    """RadioReceiverOptions::fftPlacementMethod: EarliestPeakWithBinning (1) and StrongestPeak (0)"""
    P.check_stream_vs_oracle(factory, snr, cfo, 211, 9, True, fft_placement=method)


@pytest.mark.parametrize("freqsync,cfo,snr", [(0, 2300, 20), (1, 2300, 20), (1, -1000, 14), (0, 17400, None)])
def test_other_freqsync_methods(emu, freqsync, cfo, snr):
    """RadioReceiverOptions::freqsyncMethod: GetMiddle (0) and CorrelatePRS (1) drive the coarse corrector"""
    P.check_stream_vs_oracle(factory, snr, cfo, 250, 10, True, seed=50 + freqsync, freqsync=freqsync)


@pytest.mark.parametrize("ring_extra", [44500, 77777])
def test_live_ring_that_wraps_inside_the_frames(emu, ring_extra):
    """a ring of 6 frames + 44 500 / 77 777 samples: its end falls on another OFDM symbol in every revolution -- the first symbol of a demod
    work-group's chunk (its LDS-DMA pipeline starts one symbol late) and symbols in the middle of a chunk (the pipeline is interrupted and
    picked up again); that symbol comes straight from HBM.  (Which symbols: seen by instrumenting the kernel once.)"""
    P.check_live_raw_vs_oracle(factory, "s16le", nf=20, ring_extra=ring_extra)


def test_live_ring_async_ingest(emu):
    """dabphy_stream_write_raw_async: copy + conversion on the copy stream, dabphy_process orders itself behind them"""
    P.check_live_raw_vs_oracle(factory, "u8", asynchronous=True)


def test_tii_side_path(emu):
    """TIIDecoder on the device: four ensembles with different transmitter sets, sums carried across batches"""
    P.check_tii_vs_oracle(factory)


def test_fine_corrector_interval_and_exact_paths(emu):
    P.check_fine_corrector_paths(factory)


def test_dropout_and_relock(emu):
    P.check_dropout_relock(factory)


def test_relock_after_long_lock(emu):
    P.check_relock_after_long_lock(factory)


def test_lock_lost_inside_a_replayed_batch(emu):
    P.check_lock_lost_inside_a_replayed_batch(factory)


def test_fine_corrector_on_the_edge(emu):
    P.check_fine_corrector_on_the_edge(factory)


def test_benchmark_handle_configuration_small(emu):
    """the handle bench.py opens (looping ring, coarse corrector on, pipelined synchroniser, all 18 sub-channels, superframe filter
    inside process(), 25-symbol demod chunks), at a size the execution model finishes: against the oracle on the same samples"""
    from welle_io_amd import workload
    base = workload.make_base_streams(2, workload.REC_FRAMES, seed0=0)
    # lane-per-code-word as the benchmark runs it, and (fewer steps, one ensemble checked) state-parallel: what this size gets by default on the device
    P.check_bench_config(capi, EMU_LIB, 3, 3, 1, check_ens=[0, 2], n_steps=5, demod_chunk=25, device="cpu", subs_idx=(0, 7, 17), base=base, expect_chunk=25, decode_shape=1)
    P.check_bench_config(capi, EMU_LIB, 3, 3, 1, check_ens=[1], n_steps=2, demod_chunk=25, device="cpu", subs_idx=(0, 17), base=base, expect_chunk=25, decode_shape=2)


def test_benchmark_configuration_with_the_deferred_filter(emu):
    """dabphy_set_auto_superframes(2), what bench.py's handle runs with: the superframe filter pass of a batch is queued by the NEXT
    dabphy_process beside its FFT stage; the totals arrive one batch later and sum to the same (the oracle's); with the synchroniser
    behind the decoder as in rounds 1-5 (dabphy_config.sync_early = 1; every other test runs the default: in front)"""
    from welle_io_amd import workload
    base = workload.make_base_streams(2, workload.REC_FRAMES, seed0=0)
    P.check_bench_config(capi, EMU_LIB, 3, 3, 1, check_ens=[0, 2], n_steps=5, demod_chunk=25, device="cpu", subs_idx=(0, 7, 17), base=base, expect_chunk=25, decode_shape=1,
                         deferred_filter=True, sync_early=1)


def test_heterogeneous_multiplex_small(emu):
    """bench.py's `hetero` leg (workload.HETERO_LAYOUT: 15 sub-channels, 6 protection classes incl. EEP-B and UEP, all DAB+) through the
    handle bench.py opens, at a size the execution model finishes: one fused launch for all classes and the FIC, every sub-channel's
    bytes and the superframe totals against the oracle"""
    from welle_io_amd import workload
    lib = capi.load_library(EMU_LIB)
    subchs = workload.hetero_subchannels(lib)
    assert len(subchs) == 15 and sum(s.size_cu for s in subchs) <= 864
    assert workload.subchannels_to_json(workload.subchannels_from_json(workload.subchannels_to_json(subchs))) == workload.subchannels_to_json(subchs)
    P.check_bench_config(capi, EMU_LIB, 2, 4, 1, check_ens=[0, 1], n_steps=4, demod_chunk=25, device="cpu", subs_idx=tuple(range(15)),
                         base=workload.make_base_streams(2, workload.REC_FRAMES, seed0=50, subchs=subchs), expect_chunk=25, decode_shape=1)


@pytest.mark.parametrize("fmt", ["u8", "s8", "s16le", "s16be"])
def test_ingest_vs_reference_crawfile(emu, fmt, tmp_path):
    """k_ingest pinned to the real CRAWFile::convertSamples (input/raw_file.cpp:324-366, compiled into oracle/_ref)"""
    import refapi as R
    if not R.have_ref():
        pytest.skip("oracle/_ref not built")
    P.check_ingest_vs_rawfile(factory, fmt, tmp_path)


@pytest.mark.parametrize("snr,cfo,F,seed", [(4, 300, 4, 3), (3, -1000, 4, 5), (5, 2300, 6, 7), (2, 40, 4, 9)])
def test_low_snr_batches_with_coarse_corrector(emu, snr, cfo, F, seed):
    """batch mode (F > 1) with the coarse corrector enabled while the FIC decodes badly (ratio around / below 50): the corrector then
    consults the previous batch's ratio (include/dabphy.h, dabphy_process); these streams, including losses of lock inside a batch,
    either give the oracle's frames bit for bit or part from them exactly where the ratio crossed the 50 % line inside a batch"""
    P.check_stream_vs_oracle(factory, snr, cfo, 150, 21, False, F=F, seed=seed, ratio_lag_ok=True)


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_one_frame_per_call_on_a_pipelined_schedule(emu, mode):
    """ONE frame per call on a pipelined schedule: the synchroniser is one or two FRAMES ahead of the decoder, so the coarse corrector of
    frame n + 1 (n + 2) has not yet seen the FIC of frame n -- the reference consults the ratio of the previous frame before every coarse
    step (ofdm-processor.cpp:397-409).  Exact batch mode's second pass is armed for it like for deeper batches (round 6; until round 5 this
    combination kept the reported deviation): 3 dB, -1000 Hz -- two effective stale decisions from frame 6 on when two frames ahead -- must
    equal the oracle frame for frame, every FIB, corrector, soft bit, null symbol and MSC byte, and report no lag with an effect."""
    logs = P.check_exact_batch(factory, 3, -1000, 1, 5, pipeline_sync=mode, nf=21)
    if mode == 3:
        assert logs[0]["replayed"] >= 1, logs[0]["replayed"]


def test_dropout_in_batch_mode(emu):
    P.check_dropout_batch(factory)


def test_mixed_protection_classes_fused_decode(emu):
    """16 frames per call = 64 CIFs per sub-channel: every class (EEP A/B, UEP, 8 .. 384 kbit/s) takes the fused kernel's 96-row build -- the MSC
    gather inside the Viterbi kernel (k_viterbi_fused: LDS window ring fed by LDS-DMA, per-step descriptors from the depuncturing map)"""
    P.check_mixed_ensemble(factory_lane_per_codeword, F=16, nf=36, expect_fused=True)


def test_receiver_options_at_run_time(emu):
    P.check_runtime_options(factory)


@pytest.mark.parametrize("pipeline", [False, 1, 2])
def test_wide_synchroniser_pass(emu, pipeline):
    P.check_wide_sync(factory, pipeline_sync=pipeline)


@pytest.mark.parametrize("snr,cfo,F,seed,pipeline,replay", [(3, -1000, 4, 5, False, True), (3, -1000, 4, 5, 1, True), (3, -1000, 4, 5, 3, None),
                                                            (2, 40, 4, 9, 2, None), (4, 17400, 5, 13, 1, None)])      # (the device suite runs four more)
def test_exact_batch_mode(emu, snr, cfo, F, seed, pipeline, replay):
    """Exact batch mode (the default): the low-SNR batch streams again, now required to equal the oracle frame for frame without any
    tolerance (a batch whose stale coarse-corrector decision can have mattered is put back and decoded a second time with the
    reference's per-frame FIC-ratio feedback); the first two are known to need that second pass"""
    P.check_exact_batch(factory, snr, cfo, F, seed, pipeline_sync=pipeline, expect_replay=replay)


def test_exact_batch_mode_state_parallel(emu):
    """the replay's frame-by-frame FIC decode through the one-class state-parallel launch (frame selector in the launch arguments), and the
    batch's own decode state-parallel too: the stream that is known to need the second pass"""
    P.check_exact_batch(factory_state_parallel, 3, -1000, 4, 5, pipeline_sync=False, expect_replay=True)


@pytest.mark.parametrize("chunk", [1, 2])
def test_exact_batch_mode_with_short_demod_chunks(emu, chunk):
    """the replay demodulates only the chunks that hold the FIC symbols 1 .. 3 of a frame: with one or two symbols per work-group that
    is three or two chunks (a replay that took only the first one read stale soft bits: advisor, round 2); the batch must still equal
    the oracle frame for frame and must have been decoded twice"""
    P.check_exact_batch(lambda **kw: factory(demod_chunk=chunk, **kw), 3, -1000, 4, 5, pipeline_sync=False, expect_replay=True)


def test_superframes_through_a_replayed_batch(emu):
    """exact batch mode with the superframe filter: a 3.5 dB stream in which one batch has to be decoded a second time -- the filter's
    windows are put back with the rest of the state -- gives the oracle's superframe events, corrected superframes and totals whether
    the filter is called per sub-channel, for all of them after the batch, or rides inside dabphy_process"""
    st = {}
    P.check_superframes_vs_oracle(factory, F=3, nf=22, snr_db=3.5, seed=10, B=1, damage=True, cfo=40, stats=st)
    assert st["replayed"] >= 1 and st["replayed_auto_0"] >= 1 and st["replayed_auto_1"] >= 1, st


@pytest.mark.parametrize("snr,cfo,F,pipeline", [(4, -1000, 4, False)])                                           # (the device suite also runs a pipelined case with three replays)
def test_tii_through_replayed_batches(emu, snr, cfo, F, pipeline):
    """exact batch mode with the TII side path: at 3.5-4 dB batches have to be decoded a second time (the TII sums are put back with the
    rest of the state); the measurements still equal the TIIDecoder restatement fed by the oracle receiver"""
    st = {}
    P.check_tii_vs_oracle(factory, F=F, nf=21, snr_db=snr, cfo=cfo, pipeline_sync=pipeline, stats=st, counts=False)
    assert st["replayed"] >= 1, st


@pytest.mark.parametrize("snr,cfo,seed,F", [(3, 300, 17, 3), (20, 37, 9, 4)])
def test_live_ring_in_batches(emu, snr, cfo, seed, F):
    """a live ring (s16 samples written as they arrive) decoded F frames per call: slots without samples inside the batches, the wide
    synchroniser pass and its serial fall-back, at 3 dB a loss of lock with re-acquisition -- FIBs and MSC bytes = the oracle's"""
    st = {}
    P.check_live_batch_vs_oracle(factory, F=F, snr_db=snr, cfo=cfo, seed=seed, stats=st)
    assert st["wide"][0][0] >= 1, st


def test_live_ring_shorter_than_the_lock(emu):
    """3 dB, lock lost after 12 frames in a live ring of 13: the samples sLevel would have to be replayed from are gone, the two
    bracketing replays start from 0 and from the bound of converted s16 samples (2.125, not 3e38) and meet within the history that
    is left: the re-acquisition is certified and lands where the reference's does"""
    st = {}
    P.check_live_batch_vs_oracle(factory, F=3, snr_db=3, cfo=300, seed=17, stats=st, ring_frames=13)
    assert st["lost"] >= 1 and st["relock_inexact"] == 0, st


@pytest.mark.parametrize("pipeline", [False, 3])
def test_exact_batch_mode_with_different_ensembles(emu, pipeline):
    """one ensemble of three makes a batch be decoded twice: all three must still equal their own oracle runs"""
    P.check_exact_batch_mixed(factory, pipeline_sync=pipeline)


def test_independent_ensembles_in_one_batch(emu):
    """five ensembles, five different multiplexes, each with its own sub-channel selection (dabphy_set_subchannels_ensemble): every
    selected sub-channel of every ensemble, FIBs and superframe totals against the oracle (the device twin runs 256 x 32)"""
    from conftest import EMU_LIB as lib
    P.check_mixed_layouts(capi, lib, 5, 4, check_ens=[0, 1, 2, 3, 4], n_steps=3, device="cpu", decode_shape=1, rec_frames=20)


def test_service_added_and_removed_in_mid_stream(emu):
    """dabphy_set_subchannels_ensemble between batches: the services that keep playing (in the changed ensemble and in the other one)
    deliver the uninterrupted stream's bytes and superframe events; the added one starts like a fresh DabAudio"""
    P.check_service_changes_in_mid_stream(factory)


@pytest.mark.parametrize("mode", [1, 3])
def test_service_changes_while_the_synchroniser_runs_ahead(emu, mode):
    """the same with the pipelined schedules (the next one or two batches are synchronised while this one is decoded): a selection change
    between two dabphy_process calls applies to the batch decoded next, whatever has been synchronised ahead"""
    P.check_service_changes_in_mid_stream(factory, pipeline_sync=mode)


@pytest.mark.parametrize("warm,three", [(0, False), (1, False), (0, True), (4, True)])
def test_block_parallel_traceback_with_wrong_guesses(emu, monkeypatch, warm, three):
    """k_traceback_sp2 (lane = code word, wave = a stretch of the code word walked from a guessed state and CHECKED against the exit state of
    the stretch above).  warm = the blocks of 30 steps a stretch's walk runs in over before its own: the product's 4 (every other
    state-parallel test: guesses are almost always right), here 1 (some are wrong) and 0 (every guess is wrong: every stretch is walked
    again from the true state, the rounds cascade down the code word) -- the bytes are the serial walk's in every case.  Mixed
    protection classes incl. 9216-bit code words (four stretches), 64 kbit/s ones (four) and the FIC (three).  three: the build with three
    waves per work-group, which launches of more than 512 groups take on the device (DABPHY_SP2_TB_RESIDENT = 0 forces it here)."""
    monkeypatch.setenv("DABPHY_SP2_TB_WARM", str(warm))
    if three:
        monkeypatch.setenv("DABPHY_SP2_TB_RESIDENT", "0")
    P.check_mixed_ensemble(factory_state_parallel, F=3, nf=9, expect_fused=True)


def test_service_changes_with_two_code_words_per_wavefront(emu):
    """services changing in mid-stream through k_viterbi_sp2 + k_traceback_sp2 (decode_shape = 2: what batches of 1 024 .. 40 960 code words
    take on the device)"""
    P.check_service_changes_in_mid_stream(factory_state_parallel)
