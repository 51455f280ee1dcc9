"""Pins the oracle (oracle/dabphy_oracle.c) to the REAL reference compiled by oracle/Makefile into oracle/_ref.
Skipped where the prebuilt reference libraries are absent (they are built in the authoring container, where
/root/reference exists, and travel to the GPU box as binaries); tests/test_golden.py covers that case."""
import numpy as np
import pytest

import parity_cases as P
import refapi as R
from welle_io_amd import synth

pytestmark = pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built (needs /root/reference)")


def eq_bits(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_tables():
    assert eq_bits(R.ref_prs_reftable(), R.orc_prs_reftable())
    assert np.array_equal(R.ref_freq_perm(), R.orc_freq_perm())
    for i in range(24):
        assert np.array_equal(R.ref_pcodes(i), R.orc_pcodes(i))
    assert np.array_equal(R.ref_energy(np.zeros(9216, np.uint8)), R.orc_prbs(9216))


def test_fft():
    rng = np.random.RandomState(1)
    for scale in (1.0, 1e-3, 300.0):
        x = ((rng.randn(2048) + 1j * rng.randn(2048)) * scale).astype(np.complex64)
        assert eq_bits(R.ref_fft2048(x), R.orc_fft2048(x))
        assert eq_bits(R.ref_fft2048(x, True), R.orc_fft2048(x, True))


@pytest.mark.parametrize("nbits", [768, 1536, 192, 9216])
def test_viterbi(nbits):
    rng = np.random.RandomState(nbits)
    for kind in range(3):
        s = rng.randint(-128, 128, 4 * (nbits + 6)).astype(np.int8) if kind < 2 else rng.choice(np.array([-128, 127, 0], np.int8), 4 * (nbits + 6))
        assert np.array_equal(R.ref_viterbi(s, nbits), R.orc_viterbi(s, nbits))


def test_protection_profiles():
    rng = np.random.RandomState(3)
    for br in (8, 16, 32, 64, 96, 128, 192):
        for pb in (0, 1):
            if pb and br % 32:
                continue
            for lv in (1, 2, 3, 4):
                p = R.orc_prot_eep(br, pb, lv)
                s = rng.randint(-128, 128, p.n_in).astype(np.int8)
                assert np.array_equal(R.ref_eep(br, pb, lv, s), R.orc_msc_deconvolve(p, s)), (br, pb, lv)
    for br, lv in [(32, 5), (32, 1), (48, 3), (56, 2), (64, 4), (80, 1), (96, 5), (112, 3), (128, 2), (160, 4), (192, 1), (224, 3), (256, 5), (320, 2), (384, 3)]:
        p = R.orc_prot_uep(br, lv)
        s = rng.randint(-128, 128, p.n_in).astype(np.int8)
        assert np.array_equal(R.ref_uep(br, lv, s), R.orc_msc_deconvolve(p, s)), (br, lv)


def test_fic_random():
    rng = np.random.RandomState(9)
    s = rng.randint(-128, 128, 9216).astype(np.int8)
    a = R.ref_fic_decode(s); b = R.orc_fic_decode(s)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]


def test_rs():
    rng = np.random.RandomState(11)
    sf = np.zeros(120 * 8, np.uint8)
    for i in range(8):
        dta = rng.randint(0, 256, 110).astype(np.uint8)
        sf[i::8] = np.concatenate([dta, R.orc_rs_encode120(dta)])
    a = R.ref_rs_superframe(sf)
    assert a[1] == 0 and a[2] == 0 and np.array_equal(a[0], sf)
    for trial in range(40):
        e = sf.copy(); ne = rng.randint(0, 70)
        pos = rng.choice(len(e), ne, replace=False); e[pos] ^= rng.randint(1, 256, ne).astype(np.uint8)
        a = R.ref_rs_superframe(e); b = R.orc_rs_superframe(e)
        assert np.array_equal(a[0], b[0]) and a[1:] == b[1:]


def test_ofdm_decoder_soft_bits():
    """every soft bit of the reference's own OfdmDecoder (tapped through oracle/shim) == oracle"""
    import parity_cases as P
    x = synth.make_stream(3, snr_db=13, seed=5)
    frames = P.cut_frames(x, 2)
    sr, cr, _ = R.ref_ofdm_decode_frames(frames)
    so, co, _ = R.orc_demod_frames(frames)
    assert np.array_equal(sr, so) and eq_bits(cr, co)


@pytest.mark.parametrize("snr,cfo,delay,nf", [(25, 0, 0, 22), (13, 137, 1000, 14), (20, 2300, 0, 14), (20, -400, 333, 10), (None, 17400, 0, 8), (10, -1000, 0, 8)])
def test_receiver_end_to_end(snr, cfo, delay, nf):
    """the whole chain through the reference's public RadioReceiver vs orc_receiver_run: FIBs, MSC bytes, correctors,
    NCO-mixed null symbols, impulse responses, constellation taps and SNR reports -- all bit-identical"""
    x, tx = synth.make_stream(nf, snr_db=snr, cfo_hz=cfo, delay=delay, return_tx=True, seed=3)
    subs = [tx.subchs[0], tx.subchs[5]]
    a = R.receiver_run(x, subchs=subs)
    b = R.orc_receiver_run(x, subchs=subs)
    n = min(len(a["fib"]), len(b["fib"]))
    assert n >= 12 * (nf - 3) and np.array_equal(a["fib"][:n], b["fib"][:n])
    k = min(len(a["nul"]), b["n_frames"])
    assert k >= nf - 3
    assert np.array_equal(a["corr"][:k], b["corr"][:k])
    assert eq_bits(a["nul"][:k], b["nul"][:k])
    kk = min(len(a["cir"]), len(b["cir"]))
    assert eq_bits(a["cir"][:kk], b["cir"][:kk])
    assert eq_bits(a["con"][:k], b["con"][:k])
    assert eq_bits(a["snr"], b["snr"])
    for i in range(2):
        m = min(len(a["msc"][i]), len(b["msc"][i]))
        assert a["msc"][i][:m] == b["msc"][i][:m]
    if cfo == 0:
        # and both equal what was transmitted
        pay = b"".join(tx.payload_log[subs[0].subch_id])
        assert a["msc"][0] in pay and len(a["msc"][0]) > 0


@pytest.mark.parametrize("bitrate,seed", [(64, 1), (32, 2), (128, 3), (8, 4)])
def test_superframe_filter_equals_reference(oracle_built, bitrate, seed):
    """SuperframeFilter::Feed (sliding 5-frame window, RS, Fire-code sync, AU CRCs) on a stream that starts mid-superframe,
    carries byte errors within and beyond the RS capacity, a broken AU and a header hit"""
    rng = np.random.RandomState(seed)
    fb = 3 * bitrate
    sfs = [synth.make_superframe(bitrate, rng) for _ in range(9)]
    sfs[2][7 * (bitrate // 8) + 1] ^= 0x55                                  # one correctable symbol
    for k in range(4):
        sfs[3][(20 + k) * (bitrate // 8)] ^= 0xA0 + k                        # 4 symbols in codeword 0
    a = sfs[4]; a[40] ^= 1; a[40 + bitrate // 8] ^= 2                          # errors in two codewords
    for k in range(7):
        sfs[5][(30 + 3 * k) * (bitrate // 8) + (1 % (bitrate // 8))] ^= 0x11 * (k + 1)   # beyond capacity: uncorrectable or miscorrected
    stream = np.concatenate(sfs).reshape(-1, fb)[3:]                          # start 3 frames into a superframe
    stream = np.concatenate([stream[:23], stream[24:]])                       # drop one frame: sync lost and found again
    eo, so = R.orc_superframe_run(stream)
    er, sr = R.ref_superframe_run(stream)
    assert eo == er
    assert len(so) == len(sr) and all(np.array_equal(x, y) for x, y in zip(so, sr))
    assert sum(e[3] for e in eo) >= 5 and any(e[2] for e in eo)


@pytest.mark.parametrize("net", range(len(P.TII_NETWORKS)))
@pytest.mark.parametrize("snr", [25, 8])
def test_tii_decoder_equals_reference(oracle_built, net, snr):
    """TIIDecoder::run + analyse_phase (the real class, one pair at a time) vs the restatement: same comb/pattern pairs, same
    frames, same delay (= same tie-break order of the unordered_map) and same float error, over three report cycles"""
    x = synth.make_stream(17, snr_db=snr, seed=50 + net, noise_seed=3 * net + snr, tii=P.TII_NETWORKS[net])
    nul, prs = P.tii_pairs(x, 16, early=60 + 20 * net)
    eo = R.orc_tii_run(nul, prs)
    er = R.ref_tii_run(nul, prs)
    assert eo == er
    if P.TII_NETWORKS[net] is None:
        assert eo == []
    else:
        assert len(eo) >= 2
        if snr >= 20:                                            # (at 8 dB the reference itself reports neighbouring patterns too)
            assert {e[1:3] for e in eo} <= {t[:2] for t in P.TII_NETWORKS[net]}


def test_tii_decoder_noise_and_garbage(oracle_built):
    """noise-only NULL symbols, a NULL symbol that lights every carrier (>= 10 likely pairs: the frame is skipped), zeros"""
    rng = np.random.RandomState(5)
    x = synth.make_stream(8, snr_db=15, seed=9, tii=[(7, 33, 0, 1.0)])
    nul, prs = P.tii_pairs(x, 7)
    nul[1] = (rng.randn(2656) + 1j * rng.randn(2656)).astype(np.complex64) * 0.05
    nul[2] = 0
    nul[3] = prs[3][np.arange(2656) % 2048] * 3.0            # a PRS-like NULL: all pairs over the threshold
    nul2, prs2 = np.concatenate([nul] * 3), np.concatenate([prs] * 3)
    eo, det = R.orc_tii_run(nul2, prs2, want_detect=True)
    assert eo == R.ref_tii_run(nul2, prs2) and len(eo) >= 2
    assert det[3].sum() > 100 and det[2].sum() == 0


@pytest.mark.parametrize("method", [0, 1])
def test_receiver_other_fft_placement_methods(oracle_built, method):
    """end to end through RadioReceiver with fftPlacementMethod = StrongestPeak / EarliestPeakWithBinning"""
    x, tx = synth.make_stream(9, snr_db=14, cfo_hz=60, delay=400, return_tx=True, seed=44)
    subs = [tx.subchs[3]]
    a = R.receiver_run(x, subchs=subs, fft_placement=method)
    b = R.orc_receiver_run(x, subchs=subs, fft_placement=method)
    n = min(len(a["fib"]), len(b["fib"]))
    assert n >= 12 * 6 and np.array_equal(a["fib"][:n], b["fib"][:n])
    k = min(len(a["cir"]), len(b["cir"]))
    assert np.array_equal(a["cir"][:k].view(np.uint32), b["cir"][:k].view(np.uint32))
    m = min(len(a["msc"][0]), len(b["msc"][0]))
    assert m > 0 and a["msc"][0][:m] == bytes(b["msc"][0])[:m]


@pytest.mark.parametrize("freqsync,cfo,snr", [(0, 2300, 20), (1, 2300, 20), (1, -1000, 14), (0, 17400, None), (1, 60, 9), (0, -400, 12)])
def test_receiver_other_freqsync_methods(oracle_built, freqsync, cfo, snr):
    """RadioReceiverOptions::freqsyncMethod GetMiddle (0) / CorrelatePRS (1): the coarse corrector they drive while the FIC does
    not decode, end to end through RadioReceiver -- correctors frame by frame, FIBs, MSC bytes"""
    x, tx = synth.make_stream(10, snr_db=snr, cfo_hz=cfo, delay=250, return_tx=True, seed=50 + freqsync)
    subs = [tx.subchs[7]]
    a = R.receiver_run(x, subchs=subs, freqsync=freqsync)
    b = R.orc_receiver_run(x, subchs=subs, freqsync=freqsync)
    k = min(len(a["corr"]), len(b["corr"]))
    assert k >= 5 and np.array_equal(a["corr"][:k], b["corr"][:k]), (a["corr"][:k].T, b["corr"][:k].T)
    n = min(len(a["fib"]), len(b["fib"]))
    assert np.array_equal(a["fib"][:n], b["fib"][:n])
    m = min(len(a["msc"][0]), len(b["msc"][0]))
    assert a["msc"][0][:m] == bytes(b["msc"][0])[:m]


def test_receiver_short_form_subchannels(oracle_built):
    """UEP (short-form) sub-channels through the reference's RadioReceiver -- MscHandler -> DabAudio -> UEPProtection::deconvolve --
    vs the oracle's streaming receiver: 80 kbit/s level 1 (the table row with PI2 = 7) and 384 kbit/s level 5 (9216-bit code
    words), next to an EEP one; the .msc dumps agree and equal the transmitted payload"""
    subchs = [synth.SubchannelCfg(1, 0, 64, False, 3, dabplus=False)]
    cu = subchs[0].size_cu
    for sid, br, lvl in ((8, 80, 1), (9, 384, 5)):
        sc = R.uep_subchannel(synth, sid, cu, br, lvl); subchs.append(sc); cu += sc.size_cu
    x, tx = synth.make_stream(10, subchs=subchs, snr_db=14, cfo_hz=33, delay=77, return_tx=True, seed=17)
    a = R.receiver_run(x, subchs=subchs)
    b = R.orc_receiver_run(x, subchs=subchs)
    for i, sc in enumerate(subchs):
        m = min(len(a["msc"][i]), len(b["msc"][i]))
        assert m >= 8 * sc.frame_bytes, (i, len(a["msc"][i]), len(b["msc"][i]))
        assert a["msc"][i][:m] == b["msc"][i][:m], "sub-channel %d (%d kbit/s)" % (sc.subch_id, sc.bitrate)
        assert a["msc"][i][:m] in b"".join(tx.payload_log[sc.subch_id])
