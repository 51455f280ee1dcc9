"""CPU suite: the whole streaming receiver (acquisition, PRS sync, coarse/fine tracking, demod, FIC, MSC with time
de-interleaving) compiled for tests/hipemu, against orc_receiver_run -- which tests/test_oracle_vs_ref.py pins to
the real reference end to end."""
import pytest

import parity_cases as P
from conftest import EMU_LIB
from welle_io_amd import capi


def factory(**kw):
    return capi.DabPhy(lib_path=EMU_LIB, **kw)


@pytest.mark.parametrize("snr,cfo,delay,nf,lockstep", [(25, 0, 0, 10, False), (13, 137, 1000, 8, True), (20, 2300, 0, 8, True), (20, -400, 333, 8, True)])
def test_stream(emu, snr, cfo, delay, nf, lockstep):
    P.check_stream_vs_oracle(factory, snr, cfo, delay, nf, lockstep)


def test_two_ensembles_in_lock_step(emu):
    P.check_stream_vs_oracle(factory, 18, 0, 500, 7, False, B=2, F=2)


def test_big_batch_tiled_gather(emu):
    """F = 20 (80 CIFs per sub-channel and batch): exercises the LDS-tiled MSC gather incl. groups straddling two sub-channels"""
    P.check_stream_vs_oracle(factory, 14, 30, 200, 43, False, F=20)


@pytest.mark.parametrize("mode", [1, 2])
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


def test_mixed_protection_classes(emu):
    P.check_mixed_ensemble(factory)


@pytest.mark.parametrize("method,snr,cfo", [(1, 15, 90), (1, None, -300), (0, 12, 40)])
def test_other_fft_placement_methods(emu, method, snr, cfo):
    """RadioReceiverOptions::fftPlacementMethod: EarliestPeakWithBinning (1) and StrongestPeak (0)"""
    P.check_stream_vs_oracle(factory, snr, cfo, 211, 9, True, fft_placement=method)


@pytest.mark.parametrize("freqsync,cfo,snr", [(0, 2300, 20), (1, 2300, 20), (1, -1000, 14), (0, 17400, None)])
def test_other_freqsync_methods(emu, freqsync, cfo, snr):
    """RadioReceiverOptions::freqsyncMethod: GetMiddle (0) and CorrelatePRS (1) drive the coarse corrector"""
    P.check_stream_vs_oracle(factory, snr, cfo, 250, 10, True, seed=50 + freqsync, freqsync=freqsync)


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


def test_fine_corrector_on_the_edge(emu):
    P.check_fine_corrector_on_the_edge(factory)
