"""INTEGRATION.md level 2 as a BUILD (BASELINE config 2: "2048-pt FFT + DQPSK HIP path, Viterbi still on CPU"): the reference backend
compiled from its own unmodified sources with exactly ONE file replaced by a seam binding from welle.io_amd/host/seams/
  l2a  ofdm-decoder.cpp -> ofdm_decoder_seam.cpp   OfdmDecoder::pushAllSymbols -> dabphy_demod_frames   (ofdm-decoder.cpp:132-139)
  l2b  viterbi.cpp      -> viterbi_seam.cpp        Viterbi::deconvolve         -> dabphy_viterbi_batch  (viterbi.cpp:227-245)
(oracle/Makefile `level2`), driven through the same recording harness as the pure reference build: every callback the reference's
RadioReceiver makes -- FIBs + CRC flags, SNR reports, constellation points, impulse responses, null symbols -- and every sub-channel's
dump must be the reference build's.  CPU: the kernels in the tests/hipemu execution model; -m gpu: the real library, and welle-cli
itself built both ways decoding a RAW u8 file."""
import os

import numpy as np
import pytest

import refapi as R
from test_host_mirror import compare_runs
from welle_io_amd import synth


def _have(variant, backend):
    return R.have_ref() and os.path.exists(R.level2_lib(variant, backend))


def check_seam_build(variant, backend, snr=14, cfo=137, delay=700, nf=12):
    x, tx = synth.make_stream(nf, snr_db=snr, cfo_hz=cfo, delay=delay, return_tx=True, seed=5)
    subs = [tx.subchs[2], tx.subchs[11]]
    a = R.receiver_run(x, subchs=subs)
    b = R.receiver_run(x, subchs=subs, lib=R.level2_lib(variant, backend))
    compare_runs(a, b, len(subs))
    # (the two builds share everything above the seam: the same number of frames, SNR reports and dump bytes, not only a common prefix)
    assert len(a["fib"]) == len(b["fib"]) and len(a["snr"]) == len(b["snr"]) and len(a["snr"]) >= 1
    assert np.array_equal(a["snr"], b["snr"]), "onSNR values differ"          # same libm-free arithmetic up to log10: equal here, 1e-5 by contract
    assert [len(m) for m in a["msc"]] == [len(m) for m in b["msc"]]


@pytest.mark.skipif(not _have("a", "emu"), reason="oracle/_ref level-2 builds missing (need /root/reference)")
def test_ofdm_decoder_seam_build(emu):
    check_seam_build("a", "emu")


@pytest.mark.skipif(not _have("b", "emu"), reason="oracle/_ref level-2 builds missing (need /root/reference)")
def test_viterbi_seam_build(emu):
    check_seam_build("b", "emu")


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["a", "b"])
def test_seam_builds_on_the_device(gpu, variant):
    assert _have(variant, "hip"), "oracle/_ref/libwelle_l2%s_hip.so must travel with the snapshot" % variant
    check_seam_build(variant, "hip", nf=20)
    check_seam_build(variant, "hip", snr=22, cfo=0, delay=0, nf=14)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["b"])         # (l2a: test_seam_builds_on_the_device[a] and bench.py's facade.level2)
def test_welle_cli_with_one_seam_on_the_device(gpu, variant, tmp_path):
    """welle-cli -f <RAW u8 IQ file> -D built from the reference's sources with one file replaced by the seam binding: dump.fic and every
    service's .msc dump equal the reference build's"""
    from test_welle_cli import REF_DIR, _compare_runs
    binary = os.path.join(REF_DIR, "welle-cli-l2%s-hip" % variant)
    assert os.path.exists(binary), "%s must travel with the snapshot" % binary
    _compare_runs(tmp_path, binary, 4.0, 4.0)
