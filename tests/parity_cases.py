"""Parity checks shared by the CPU (hipemu) and GPU runs: product C ABI vs the oracle on the same seeded inputs.
Bit-exact everywhere (integer/byte outputs and the float constellation taps); the only tolerance is on the SNR
report (north star: floats within 1e-5), stated where it is used."""
import numpy as np

import refapi as R
from welle_io_amd import synth


def cut_frames(x, n, early=100):
    """[PRS useful part][75 symbols] per frame from a clean synthetic stream; the FFT window starts `early`
    samples inside the cyclic prefix, as the reference's ThresholdBeforePeak placement does."""
    frames = np.zeros((n, 2048 + 75 * 2552), np.complex64)
    for f in range(n):
        p0 = f * 196608 + 2656 + 504 - early
        frames[f] = x[p0:p0 + 2048 + 75 * 2552]
    return frames


def check_viterbi(d, nbits, n, seed, kind="uniform"):
    rng = np.random.RandomState(seed)
    if kind == "uniform":
        s = rng.randint(-128, 128, (n, 4 * (nbits + 6))).astype(np.int8)
    elif kind == "extreme":
        s = rng.choice(np.array([-128, -127, 127, 0], np.int8), (n, 4 * (nbits + 6)))
    elif kind == "zeros":
        s = np.zeros((n, 4 * (nbits + 6)), np.int8)
    else:  # valid codewords + noise
        s = np.zeros((n, 4 * (nbits + 6)), np.int8)
        for i in range(n):
            bits = rng.randint(0, 2, nbits).astype(np.uint8)
            c = synth.conv_encode(bits).astype(np.float64)
            v = (2 * c - 1) * 40 + rng.randn(len(c)) * 35
            s[i] = np.clip(np.round(v), -127, 127).astype(np.int8)
    out = d.viterbi_batch(s, nbits)
    ref = np.stack([np.packbits(R.orc_viterbi(s[i], nbits)) for i in range(n)])
    assert np.array_equal(out, ref), "viterbi nbits=%d kind=%s: %d differing bytes" % (nbits, kind, (out != ref).sum())


def check_msc_deconvolve(d, kind, bitrate, a, b, n, seed):
    rng = np.random.RandomState(seed)
    if kind == "eep":
        p = d.protection_eep(bitrate, a, b); po = R.orc_prot_eep(bitrate, a, b)
    else:
        p = d.protection_uep(bitrate, a); po = R.orc_prot_uep(bitrate, a)
    nin = d.protection_input_bits(p)
    assert nin == po.n_in
    s = rng.randint(-128, 128, (n, nin)).astype(np.int8)
    out = d.msc_deconvolve(p, s)
    ref = np.stack([np.packbits(R.orc_msc_deconvolve(po, s[i]) ^ R.orc_prbs(p.nbits)) for i in range(n)])
    assert np.array_equal(out, ref)


def check_fic(d, n_frames, snr_db, seed):
    x = synth.make_stream(n_frames + 1, snr_db=snr_db, seed=seed)
    frames = cut_frames(x, n_frames)
    soft, _, _ = R.orc_demod_frames(frames)
    fic_soft = soft[:, :3].reshape(n_frames, 9216)
    fib, ok, ratio = d.fic_decode(fic_soft)
    r = 0
    for f in range(n_frames):
        b, k, r10 = R.orc_fic_decode(fic_soft[f], r)
        r = r10 // 10
        assert np.array_equal(ok[f], k)
        assert np.array_equal(fib[f], np.packbits(b, axis=1))
    return int(ok.sum())


def check_fic_arbitrary_int8(d, n_frames=3, seed=17):
    """dabphy_fic_decode takes the caller's soft bits: any int8, -128 included -- which the reference's symbol mapping clamps like -127
    (viterbi.cpp:233-236).  Random bytes (no FIB passes its CRC: the decoded BYTES are the check), either Viterbi kernel"""
    rng = np.random.RandomState(seed)
    s = rng.randint(-128, 128, (n_frames, 9216)).astype(np.int8)
    s[:, ::7] = -128
    fib, ok, ratio = d.fic_decode(s)
    r = 0
    for f in range(n_frames):
        b, k, r10 = R.orc_fic_decode(s[f], r)
        r = r10 // 10
        assert np.array_equal(ok[f], k)
        assert np.array_equal(fib[f], np.packbits(b, axis=1)), "frame %d: FIB bytes of arbitrary int8 input differ" % f


def check_demod(d, n_frames, snr_db, seed, early=100):
    x = synth.make_stream(n_frames + 1, snr_db=snr_db, seed=seed)
    frames = cut_frames(x, n_frames, early)
    d.reset()                      # fresh OfdmDecoder state (SNR filter + report counter), like the oracle call below
    soft, con, snr = d.demod_frames(frames)
    so, co, sn = R.orc_demod_frames(frames)
    assert np.array_equal(soft, so), "%d of %d soft bits differ" % ((soft != so).sum(), soft.size)
    assert soft.min() >= -127                 # the demapper cannot produce -128 (the fused MSC decode relies on it)
    assert np.array_equal(con.view(np.uint32), co.view(np.uint32)), "constellation points differ"
    rep = snr[~np.isnan(snr)]
    assert len(rep) == len(sn)
    if len(sn):
        assert np.allclose(rep, sn, rtol=1e-5, atol=1e-5)      # north-star tolerance for SNR floats
    return soft


def check_demod_degenerate(d, seed=21):
    """magnitudes outside the range where the demapper's fast 127/x applies, zeroed symbols, and NaN / inf samples: the
    float -> int8 corner cases of ofdm-decoder.cpp:208-212 as an x86-64 build of the reference resolves them"""
    x = synth.make_stream(3, snr_db=20, seed=seed)
    base = cut_frames(x, 2, 100)
    cases = []
    for scale in (1e-30, 1e-22, 1e-12, 1e14, 3e18):
        cases.append((base * np.complex64(scale)).astype(np.complex64))
    z = base.copy(); z[0, 2048 + 5 * 2552:2048 + 7 * 2552] = 0; z[1, :4000] = 0
    cases.append(z)
    w = base.copy(); w[0, 2048 + 9 * 2552 + 700] = np.nan; w[1, 2048 + 30 * 2552 + 900] = np.inf
    cases.append(w)
    for frames in cases:
        d.reset()
        soft, con, snr = d.demod_frames(frames)
        with np.errstate(all="ignore"):
            so, co, sn = R.orc_demod_frames(frames)
        assert np.array_equal(soft, so), "%d of %d soft bits differ" % ((soft != so).sum(), soft.size)


def dev_prot(d, s):
    """device protection record of a synth.SubchannelCfg (dabphy_protection_eep / _uep)"""
    return d.protection_uep(s.bitrate, s.level) if getattr(s, "uep", None) is not None else d.protection_eep(s.bitrate, s.profile_b, s.level)


def run_stream(d_factory, x, subs, F, n_frames_total, disable_coarse=False, B=1, pipeline_sync=False, con=True, fft_placement=2, freqsync=2, stage_log=None, serial_sync=False, exact_batch=True,
               max_frames=None, log_ens=None):
    """drive the streaming receiver over the same stream for B ensembles; returns per-ensemble logs of valid frames (max_frames: the
    handle's batch depth when it is to differ from the F frames per call; log_ens: log only these ensembles)"""
    from welle_io_amd import capi  # noqa: F401
    d = d_factory(n_ensembles=B, max_frames=max_frames or F, disable_coarse=disable_coarse, pipeline_sync=pipeline_sync, want_constellation=con, fft_placement=fft_placement, freqsync_method=freqsync, serial_sync=serial_sync, exact_batch=exact_batch)
    try:
        d.stream_upload(np.tile(np.asarray(x, np.complex64), (B, 1)))
        d.set_subchannels([(s.subch_id, s.start_cu, s.size_cu, dev_prot(d, s)) for s in subs])
        if stage_log is not None:
            d.set_profiling(True)
        logs = [dict(fib=[], ok=[], info=[], con=[], soft=[], nul=[], msc=[[] for _ in subs]) for _ in range(B)]
        done = 0
        while done < n_frames_total + 2 * F:
            d.process(F)
            info = d.frame_info(); fb, ok = d.fibs(); cn = d.constellation() if con else np.zeros((B, F, 1200), np.complex64)
            nl = d.null_symbols()
            mscs = [d.msc(i) for i in range(len(subs))]
            for b in (range(B) if log_ens is None else log_ens):
                nv = 0
                for f in range(F):
                    if info[b, f]["valid"] == 1:
                        L = logs[b]
                        L["fib"].append(fb[b, f]); L["ok"].append(ok[b, f]); L["info"].append(info[b, f]); L["con"].append(cn[b, f]); L["nul"].append(nl[b, f])
                        if b == 0:
                            L["soft"].append(d.soft_bits(b, f))
                        nv += 1
                for i in range(len(subs)):
                    m, fv = mscs[i]
                    logs[b]["msc"][i].append(m[b, fv[b]:4 * nv].tobytes())
                assert len(subs) == 0 or d.msc_rows[b] == 4 * nv
            if stage_log is not None:
                stage_log["times"] = d.stage_times()
            if (info["valid"] == 0).all():              # starved: the stream has ended (a failed window search is valid = 3 and goes on)
                break
            done += F
        stale, first = d.ratio_lag()
        eff, first_eff = d.ratio_lag_effect()
        wf, passes, fallbacks = d.wide_sync_stats()
        cf = d.find_chain_stats()
        logs[0]["osc"] = d.osc_stats()
        logs[0]["replayed"] = d.replayed_batches()
        for b in range(B):
            logs[b]["ratio_lag"] = (int(stale[b]), int(first[b]))
            logs[b]["ratio_lag_effect"] = (int(eff[b]), int(first_eff[b]))
            logs[b]["wide"] = (int(wf[b]), passes, fallbacks); logs[b]["chain_frames"] = int(cf[b])
        return logs
    finally:
        d.close()


def fic_ratio_before(ok_flags):
    """FicHandler's saturating success counter (fic-handler.cpp:219-229) x 10 as it stands BEFORE each frame: what
    OFDMProcessor::run consults for that frame's coarse corrector (ofdm-processor.cpp:397)"""
    r = 0; out = []
    for frame in ok_flags:
        out.append(10 * r)
        for k in frame:
            r = min(10, r + 1) if k else max(0, r - 1)
    return out


def check_stream_vs_oracle(d_factory, snr_db, cfo, delay, nf, lockstep, B=1, seed=3, F=4, pipeline_sync=False, disable_coarse=False, con=True, fft_placement=2, freqsync=2,
                           ratio_lag_ok=False, serial_sync=False, exact_batch=None, channel=None):
    """ratio_lag_ok: batch mode's documented deviation (include/dabphy.h, dabphy_process) is tolerated and PINNED: the frames must
    equal the oracle's up to the first frame before which the FIC ratio crossed the 50 % line within the last F (2F when pipelined)
    frames -- only there may a batch have consulted a stale ratio; what comes before is compared bit for bit, returns that frame.
    exact_batch: None = the library's default (replay on) unless ratio_lag_ok asks for the report-only mode it pins"""
    if exact_batch is None:
        exact_batch = not ratio_lag_ok
    x, tx = synth.make_stream(nf, snr_db=snr_db, cfo_hz=cfo, delay=delay, return_tx=True, seed=seed, channel=channel)
    subs = [tx.subchs[0], tx.subchs[5], tx.subchs[9]]
    o = R.orc_receiver_run(x, subchs=subs, want_soft=True, disable_coarse=disable_coarse, fft_placement=fft_placement, freqsync=freqsync)
    logs = run_stream(d_factory, x, subs, 1 if lockstep else F, o["n_frames"], B=B, pipeline_sync=pipeline_sync, disable_coarse=disable_coarse, con=con, fft_placement=fft_placement, freqsync=freqsync, serial_sync=serial_sync, exact_batch=exact_batch)
    for b in range(B):
        L = logs[b]
        n = min(len(L["fib"]), len(o["fib"]) // 12)
        if ratio_lag_ok:
            ofib = o["fib"][:12 * n].reshape(n, 12, 33)
            inf = np.array(L["info"][:n])
            same = [np.array_equal(L["ok"][k], ofib[k, :, 0]) and np.array_equal(L["fib"][k], ofib[k, :, 1:]) and
                    (int(inf["fine"][k]), int(inf["coarse"][k])) == tuple(int(v) for v in o["corr"][k]) for k in range(n)]
            stale, first_stale = L["ratio_lag"]
            if all(same):
                pass                                    # (a stale decision need not change anything: the corrector's step may have been zero)
            else:
                # the library knows exactly where its synchroniser consulted a stale ratio: nothing may differ before that frame
                assert stale > 0 and 0 <= first_stale <= same.index(False), "frame %d differs from the oracle, the first stale coarse decision is reported at frame %d" % (same.index(False), first_stale)
                # ... and more precisely where a stale decision can have MATTERED (it moved the corrector, or the corrector was not asked):
                # as long as the library reports none of those, batch mode IS the reference
                eff, first_eff = L["ratio_lag_effect"]
                assert eff > 0 and 0 <= first_eff <= same.index(False), "frame %d differs from the oracle, the first stale decision with an effect is reported at frame %d (%d in all)" % (same.index(False), first_eff, eff)
            if not all(same):
                k = same.index(False)
                rb = fic_ratio_before(o["fib"].reshape(-1, 12, 33)[:, :, 0])
                lag = F * (2 if pipeline_sync else 1)
                crossed = [f for f in range(max(1, k - lag), k + 1) if any((rb[f] < 50) != (rb[g] < 50) for g in range(max(0, f - lag), f))]
                assert crossed, "frame %d differs from the oracle although the FIC ratio did not cross 50 %% in the %d frames before it: %s" % (k, lag, rb[max(0, k - lag):k + 1])
                n = k                                   # everything before the tolerated divergence is compared below
                assert n >= 1, n
        else:
            assert n >= o["n_frames"] - (1 if lockstep else F) * (1 + {0: 0, 1: 1, 2: 1, 3: 2}[int(pipeline_sync)]), (n, o["n_frames"])
            if lockstep or disable_coarse:
                assert L["ratio_lag"] == (0, -1), L["ratio_lag"]      # one frame per call / no coarse corrector: never a stale decision
        ofib = o["fib"][:12 * n].reshape(n, 12, 33)
        assert np.array_equal(np.array(L["ok"][:n]), ofib[:, :, 0]), "CRC flags differ"
        assert np.array_equal(np.array(L["fib"][:n]), ofib[:, :, 1:]), "FIB bytes differ"
        inf = np.array(L["info"][:n])
        assert np.array_equal(np.stack([inf["fine"], inf["coarse"]], 1), o["corr"][:n]), "correctors differ"
        if con:
            assert np.array_equal(np.array(L["con"][:n]).view(np.uint32), o["con"][:n].view(np.uint32)), "constellation differs"
        kn = min(n, len(o["nul"]))
        assert np.array_equal(np.array(L["nul"][:kn]).view(np.uint32), o["nul"][:kn].view(np.uint32)), "null symbols differ"
        if b == 0:
            assert np.array_equal(np.array(L["soft"][:n]), o["soft"][:n]), "soft bits differ"
        rep = inf["snr"][~np.isnan(inf["snr"])]
        assert np.array_equal(rep, o["snr"][:len(rep)])      # same libm-free arithmetic up to log10: equal here, 1e-5 by contract
        for i in range(len(subs)):
            got = b"".join(L["msc"][i])
            if ratio_lag_ok:
                got = got[:max(0, 4 * n - 16) * subs[i].frame_bytes]
            assert len(got) > 0 or n < 5
            assert got == o["msc"][i][:len(got)], "MSC bytes of sub-channel %d differ" % i
    return logs, o, tx


# ---- live ring fed in raw sample formats (dabphy_stream_open / dabphy_stream_write_raw = CRAWFile::convertSamples on the device)
def raw_encode(x, fmt):
    """-> (raw array [n][2] as stored in a file of that CRAWFileFormat, cf32 samples raw_file.cpp:324-366 makes of it)"""
    iq = np.stack([x.real, x.imag], 1)
    if fmt == "u8":
        raw = np.clip(np.round(iq * 127.0) + 128, 0, 255).astype(np.uint8)
        f = ((raw.astype(np.int32) - 128).astype(np.float32) / np.float32(128.0)).astype(np.float32)
    elif fmt == "s8":
        raw = np.clip(np.round(iq * 127.0), -128, 127).astype(np.int8)
        f = (raw.astype(np.float32) / np.float32(128.0)).astype(np.float32)
    else:
        v = np.clip(np.round(iq * 20000.0), -32768, 32767).astype(np.int16)
        # the reference's "S16LE" reads (byte0 << 8) | byte1, its "S16BE" (byte1 << 8) | byte0
        raw = v.astype(">i2" if fmt == "s16le" else "<i2")
        f = v.astype(np.float32)
    return raw, (f[:, 0] + 1j * f[:, 1]).astype(np.complex64)


def check_live_raw_vs_oracle(d_factory, fmt, nf=9, seed=5, snr_db=18, cfo=-35, asynchronous=False, ring_extra=0):
    """ring_extra: the ring is 6 frames + this many samples long, so that the place where it wraps moves through the frame from revolution
    to revolution (the demod kernel takes the symbol that straddles the end straight from HBM instead of through its LDS-DMA pipeline)"""
    T_F = 196608
    x, tx = synth.make_stream(nf, snr_db=snr_db, cfo_hz=cfo, delay=300, return_tx=True, seed=seed)
    subs = [tx.subchs[2], tx.subchs[11]]
    raw, xf = raw_encode(x, fmt)
    o = R.orc_receiver_run(xf, subchs=subs)
    d = d_factory(n_ensembles=1, max_frames=1, want_constellation=False)
    try:
        ring = 6 * T_F + ring_extra
        d.stream_open(ring)
        d.set_subchannels([(s.subch_id, s.start_cu, s.size_cu, d.protection_eep(s.bitrate, s.profile_b, s.level)) for s in subs])
        fibs, oks, msc = [], [], [[] for _ in subs]
        wr = 0
        hold = [None, None]
        def feed(n):
            nonlocal wr
            n = min(n, len(raw) - wr)
            if n > 0:
                if asynchronous:                                   # two alternating host buffers, as the contract asks
                    buf = hold[len(hold) % 2] = np.ascontiguousarray(raw[wr:wr + n]).copy(); hold.append(None)
                    d.stream_write_raw_async(buf, fmt); d.stream_commit()
                else:
                    d.stream_write_raw(raw[wr:wr + n], fmt)
                wr += n
        feed(3 * T_F)
        idle = 0
        while idle < 3:
            d.process(1)
            info = d.frame_info()
            if info[0, 0]["valid"] == 1:
                fb, ok = d.fibs(); fibs.append(fb[0, 0]); oks.append(ok[0, 0]); idle = 0
                for i in range(len(subs)):
                    m, fv = d.msc(i); msc[i].append(m[0, fv[0]:4].tobytes())
            else:
                idle += 1
            room = ring - (wr - d.stream_consumed())
            feed(min(room, T_F))
        n = len(fibs)
        assert n >= o["n_frames"] - 1, (n, o["n_frames"])
        ofib = o["fib"][:12 * n].reshape(n, 12, 33)
        assert np.array_equal(np.array(oks), ofib[:, :, 0]) and np.array_equal(np.array(fibs), ofib[:, :, 1:]), "FIBs differ"
        for i in range(len(subs)):
            got = b"".join(msc[i]); want = bytes(o["msc"][i])
            assert len(got) > 0 and got == want[:len(got)], "MSC bytes of sub-channel %d differ" % i
    finally:
        d.close()


# ---- DAB+ superframe filter on the device vs the oracle's (itself pinned to the real SuperframeFilter)
def check_superframes_vs_oracle(d_factory, F=3, nf=16, snr_db=5.0, seed=12, B=2, damage=True, auto_modes=(False, True), cfo=20, stats=None, min_synced=1,
                                ensemble=None, pick=(1, 6), damage_q=(6, 8)):
    """superframes straddle the batches (12 logical frames per batch, 5 per superframe); the noise level makes the Viterbi
    output carry byte errors for Reed-Solomon to correct (no loss of lock: batch mode and the reference drop different
    frames then), and the transmitter damages some superframes beyond repair: a broken access unit, more byte errors than
    RS(120,110) corrects, a Fire-code hit that costs the synchronisation"""
    base = synth.dabplus_payload_fn(80, seed)

    def payload(sc, r):
        data = bytearray(base(sc, r))
        if damage:
            q, k = (r % 80) // 5, r % 5
            if q == damage_q[0] and k == 2: data[40] ^= 0x5A                         # inside an access unit of superframe 6: AU CRC fails after RS has nothing to say
            if q == damage_q[0] and k == 2:
                for j in range(12): data[3 + j * (sc.bitrate // 8)] ^= 0x33          # ... because codeword 3 is beyond repair
            if q == damage_q[1] and k == 0: data[0] ^= 0xFF; data[sc.bitrate // 8] ^= 0xFF; data[2 * (sc.bitrate // 8)] ^= 1; data[3 * (sc.bitrate // 8)] ^= 7
            if q == damage_q[1] and k == 0:
                for j in range(4, 10): data[j * (sc.bitrate // 8)] ^= 0x81           # header column uncorrectable -> Fire code fails -> window slides
        return bytes(data)
    # ensemble: the sub-channels of the multiplex (default: 18 x 64 kbit/s); pick: the ones the filter is checked on
    x, tx = synth.make_stream(nf, snr_db=snr_db, cfo_hz=cfo, delay=50, return_tx=True, seed=seed, payload_fn=payload, subchs=ensemble)
    subs = [tx.subchs[i] for i in pick]
    o = R.orc_receiver_run(x, subchs=subs)
    d = d_factory(n_ensembles=B, max_frames=F, want_constellation=False)
    try:
        d.stream_upload(np.tile(np.asarray(x, np.complex64), (B, 1)))
        d.set_subchannels([(s.subch_id, s.start_cu, s.size_cu, d.protection_eep(s.bitrate, s.profile_b, s.level)) for s in subs])
        got = [[[] for _ in subs] for _ in range(B)]; got_sf = [[[] for _ in subs] for _ in range(B)]
        for _ in range((nf + F - 1) // F):
            d.process(F)
            info = d.frame_info()
            if not (info["valid"] == 1).any():
                break
            for i, sc in enumerate(subs):
                ev, ne, sf = d.superframes(i, sc.bitrate)
                for b in range(B):
                    for k in range(ne[b]):
                        e = ev[b, k]
                        got[b][i].append((int(e["corrected"]), int(e["uncorrectable"]), int(e["sync"]), int(e["format"]), int(e["num_aus"]),
                                          tuple(int(v) for v in e["au_start"][:e["num_aus"] + 1]) if e["sync"] else (), int(e["au_crc_ok"])))
                        if e["sync"]:
                            got_sf[b][i].append(sf[b, e["sf_slot"]].copy())
        for i, sc in enumerate(subs):
            fb = 3 * sc.bitrate
            frames = np.frombuffer(bytes(o["msc"][i]), np.uint8)
            frames = frames[:len(frames) // fb * fb].reshape(-1, fb)
            eo, so = R.orc_superframe_run(frames)
            want = [e[1:] for e in eo]
            for b in range(B):
                n = len(got[b][i])
                assert n >= len(want) - 4 and n > 0, (n, len(want))
                assert got[b][i] == [(w[0], w[1], w[2], w[3], w[4], w[5], w[6]) for w in want[:n]], "superframe events of sub-channel %d differ" % i
                ns = len(got_sf[b][i])
                assert all(np.array_equal(got_sf[b][i][k], so[k]) for k in range(ns)), "corrected superframes differ"
                assert ns >= 1 or min_synced == 0, "no superframe synchronised"
        if stats is not None:
            stats["replayed"] = d.replayed_batches()
        # both ways through the filter were taken: batches whose attempts all synchronised were settled by the wide pass, the damaged
        # ones (and the first, which only fills the window) were walked frame by frame
        settled, tried = d.wide_superframe_stats()
        assert tried > 0 and (0 < settled < tried if damage and nf >= 16 and damage_q == (6, 8) else settled <= tried), (settled, tried)
        if stats is not None:
            stats["sf_wide"] = (settled, tried)
    finally:
        d.close()
    # the all-sub-channels variant: only totals leave the device -- launched by superframes_stats(), or by process() itself
    for auto in auto_modes:
        d = d_factory(n_ensembles=B, max_frames=F, want_constellation=False)
        try:
            d.stream_upload(np.tile(np.asarray(x, np.complex64), (B, 1)))
            d.set_subchannels([(s.subch_id, s.start_cu, s.size_cu, d.protection_eep(s.bitrate, s.profile_b, s.level)) for s in subs])
            d.set_auto_superframes(auto)
            tot = np.zeros((B, 4), np.int64)
            for _ in range((nf + F - 1) // F):
                d.process(F)
                if not (d.frame_info()["valid"] == 1).any():
                    break
                tot += d.superframes_stats()
            if stats is not None:
                stats["replayed_auto_%d" % int(auto)] = d.replayed_batches()
            for b in range(B):
                ev = [e for i in range(len(subs)) for e in got[b][i]]
                want = (sum(e[2] for e in ev), sum(e[0] for e in ev), sum(e[1] for e in ev), sum(e[4] - bin(e[6]).count("1") for e in ev if e[2]))
                assert tuple(tot[b]) == want, (auto, tuple(tot[b]), want)
        finally:
            d.close()
    return got


MIXED_CFGS = [(1, 32, False, 1), (2, 128, False, 2), (3, 64, True, 3), (4, 48, False, 4), (5, 8, False, 3), (6, 192, False, 3), (7, 32, True, 1)]


def mixed_subchannels(cfgs=MIXED_CFGS, uep=((8, 80, 1), (9, 384, 5))):
    """an ensemble whose sub-channels all differ: bit rates 8..192 kbit/s, EEP profiles A and B, levels 1..4; short-form (UEP)
    sub-channels through the streaming path: 80 kbit/s level 1 (the reference's table row with PI2 = 7, uep-protection.cpp:66) and the
    longest code word there is, 384 kbit/s (9216 bits, level 5)"""
    subchs = []; cu = 0
    for sid, br, pb, lvl in cfgs:
        sc = synth.SubchannelCfg(sid, cu, br, pb, lvl, dabplus=False); subchs.append(sc); cu += sc.size_cu
    for sid, br, lvl in uep:
        sc = R.uep_subchannel(synth, sid, cu, br, lvl); subchs.append(sc); cu += sc.size_cu
    assert cu <= 864
    return subchs


def check_mixed_ensemble(d_factory, F=4, nf=11, snr_db=12, seed=31, expect_fused=None, B=2, max_frames=None, subchs=None, check_ens=None):
    """an ensemble whose sub-channels all differ -- one Viterbi class per sub-channel, code words of 192..9216 bits, groups of 64 code
    words that straddle sub-channels and ensembles -- B copies side by side, each against the oracle.  expect_fused: whether every
    class went through the fused kernel (no separate gather stage) or none did.  max_frames: batch depth of the handle (ring slices of
    max_frames + 6 frames per ensemble) when it is to differ from the F frames per call; check_ens: the ensembles to compare (all)"""
    if subchs is None:
        subchs = mixed_subchannels()
    x, tx = synth.make_stream(nf, subchs=subchs, snr_db=snr_db, cfo_hz=-55, delay=123, return_tx=True, seed=seed)
    # (the coarse corrector is off when F is large: a first batch of 16 frames would consult the start-up FIC ratio for all of them,
    # the documented batch-mode deviation that test_low_snr_batches_with_coarse_corrector pins)
    o = R.orc_receiver_run(x, subchs=subchs, disable_coarse=F > 4)
    seen = {}
    ens = list(range(B)) if check_ens is None else list(check_ens)
    logs = run_stream(d_factory, x, subchs, F, o["n_frames"], B=B, stage_log=seen, disable_coarse=F > 4, max_frames=max_frames, log_ens=ens, con=B <= 2)
    if expect_fused is not None:
        # the fused kernel (gather inside the Viterbi kernel) leaves no separate gather stage
        assert (seen["times"]["msc_gather"] == 0.0) == expect_fused, seen["times"]
    for b in ens:
        L = logs[b]
        n = min(len(L["fib"]), len(o["fib"]) // 12)
        assert n >= o["n_frames"] - F
        ofib = o["fib"][:12 * n].reshape(n, 12, 33)
        assert np.array_equal(np.array(L["ok"][:n]), ofib[:, :, 0]) and np.array_equal(np.array(L["fib"][:n]), ofib[:, :, 1:]), "ensemble %d: FIBs differ" % b
        for i in range(len(subchs)):
            got = b"".join(L["msc"][i])
            assert len(got) > 0 and got == bytes(o["msc"][i])[:len(got)], "ensemble %d: MSC bytes of sub-channel %d (%d kbit/s) differ" % (b, i, subchs[i].bitrate)


def tii_pairs(x, n, early=100):
    """(NULL, PRS) pairs as OFDMProcessor would cut them from a clean stream with the FFT window `early` samples inside the prefix"""
    nul = np.zeros((n, 2656), np.complex64); prs = np.zeros((n, 2048), np.complex64)
    for f in range(n):
        p0 = f * 196608 + 2656 + 504 - early
        prs[f] = x[p0:p0 + 2048]
        nul[f] = x[p0 + 2048 + 75 * 2552:p0 + 2048 + 75 * 2552 + 2656]
    return nul, prs


TII_NETWORKS = [
    [(3, 17, 0, 1.0), (11, 40, 37, 0.6)],          # two transmitters, 37 samples apart
    [(23, 69, 0, 1.0)],                            # last comb, last pattern
    None,                                          # no TII in the null symbol: nothing to report
    [(0, 0, 0, 0.8), (1, 0, 12, 0.7), (5, 35, 60, 0.5)],
]


def check_tii_vs_oracle(d_factory, F=4, nf=17, snr_db=20, cfo=70, pipeline_sync=False, stats=None, counts=True):
    """TII side path over a batch of different single-frequency networks vs the restated TIIDecoder fed by the oracle receiver
    (itself pinned to the real class, test_oracle_vs_ref.py).  Measurements must agree exactly: comb, pattern, the frame that
    completed them, delay_samples and the float error."""
    B = len(TII_NETWORKS)
    xs = [synth.make_stream(nf, snr_db=snr_db, cfo_hz=cfo * (1 - b), delay=100 + 50 * b, seed=40 + b, noise_seed=7 + b, tii=TII_NETWORKS[b]) for b in range(B)]
    n = max(len(x) for x in xs)
    xs = [np.concatenate([x, np.zeros(n - len(x), np.complex64)]) for x in xs]
    want = [R.orc_receiver_run(x, tii=True) for x in xs]
    assert not counts or (len(want[0]["tii"]) >= 4 and len(want[1]["tii"]) >= 2 and len(want[2]["tii"]) == 0 and len(want[3]["tii"]) >= 4), [len(w["tii"]) for w in want]
    d = d_factory(n_ensembles=B, max_frames=F, pipeline_sync=pipeline_sync, want_constellation=False)
    try:
        d.stream_upload(np.stack(xs))
        d.set_tii(True)
        got = [[] for _ in range(B)]; nvalid = [0] * B
        for step in range((nf + F - 1) // F):
            d.process(F)
            info = d.frame_info(); ev, n = d.tii()
            if not (info["valid"] == 1).any():
                break
            for b in range(B):
                assert n[b] == len(ev[b])
                valid_before = np.concatenate([[0], np.cumsum(info[b]["valid"] == 1)])
                for e in ev[b]:
                    assert info[b, e["frame"]]["valid"] == 1
                    got[b].append((nvalid[b] + int(valid_before[e["frame"]]), int(e["comb"]), int(e["pattern"]), int(e["delay_samples"]), float(e["error"])))
                nvalid[b] += int(valid_before[-1])
        for b in range(B):
            nfr = nvalid[b]
            w = [e for e in want[b]["tii"] if e[0] < nfr]
            assert sorted(got[b]) == w, "TII measurements of ensemble %d differ:\n got  %s\n want %s" % (b, sorted(got[b]), w)
            assert nfr >= want[b]["n_frames"] - F * (2 if pipeline_sync else 1)
        if stats is not None:
            stats["replayed"] = d.replayed_batches()
    finally:
        d.close()


def check_fine_corrector_paths(d_factory):
    """k_sync_finish settles the int16 fine-corrector step by an interval test on exact sums and falls back to the reference's
    ordered float sums when the interval straddles a step: both paths must give the oracle's correctors, and both must be taken"""
    seen_fast = seen_exact = 0
    for snr, cfo, seed in ((25, 0, 3), (13, 137, 3), (14, -400, 3), (None, -1000, 4), (9, 30, 5)):
        x = synth.make_stream(14, snr_db=snr, cfo_hz=cfo, delay=300, seed=seed)
        o = R.orc_receiver_run(x)
        d = d_factory(n_ensembles=1, max_frames=4, want_constellation=False)
        try:
            d.stream_upload(x[None, :])
            corr = []
            for _ in range(4):
                d.process(4)
                corr += [(int(i["fine"]), int(i["coarse"])) for i in d.frame_info()[0] if i["valid"] == 1]
            lost, ex = d.sync_stats()
        finally:
            d.close()
        n = min(len(corr), len(o["corr"]))
        assert n >= 11 and corr[:n] == [tuple(c) for c in o["corr"][:n]], "correctors differ at snr %s cfo %s" % (snr, cfo)
        assert lost[0] == 0 and 0 <= ex[0] <= len(corr)
        seen_exact += int(ex[0]); seen_fast += len(corr) - int(ex[0])
    assert seen_fast >= 30 and seen_exact >= 5, (seen_fast, seen_exact)


def check_fine_corrector_on_the_edge(d_factory):
    """carrier offsets that leave the residual right at the fine corrector's step (a step is taken when the measured angle exceeds
    pi/50, i.e. a residual of about 10 Hz): the decision hangs on the last bits of FreqCorr -- the interval test must decline when it
    has to, and the ordered sums must give the reference's step.  (The offsets were found by scanning: two bands 0.08 Hz wide.)"""
    cfos = [10.09 + 0.01 * b for b in range(10)] + [10.59 + 0.01 * b for b in range(10)]
    B = len(cfos)
    base = synth.make_stream(9, snr_db=None, cfo_hz=0.0, delay=100, seed=60)
    n = np.arange(len(base))
    xs = [(base * np.exp(2j * np.pi * c * n / 2048000.0)).astype(np.complex64) for c in cfos]
    want = [R.orc_receiver_run(x, disable_coarse=True) for x in xs]
    d = d_factory(n_ensembles=B, max_frames=4, want_constellation=False, disable_coarse=True)
    try:
        d.stream_upload(np.stack(xs))
        got = [[] for _ in range(B)]
        for _ in range(2):
            d.process(4)
            info = d.frame_info()
            for b in range(B):
                got[b] += [(int(i["fine"]), int(i["coarse"])) for i in info[b] if i["valid"] == 1]
        lost, ex = d.sync_stats()
    finally:
        d.close()
    for b in range(B):
        k = min(len(got[b]), len(want[b]["corr"]))
        assert k >= 7 and got[b][:k] == [tuple(int(v) for v in c) for c in want[b]["corr"][:k]], (b, cfos[b], got[b][:k], want[b]["corr"][:k].tolist())
    assert (ex > 0).sum() >= 10 and (ex == 0).sum() >= 1, ex                    # the edge was really met, and not everywhere


def check_relock_after_long_lock(d_factory, n_locked=70, F=1):       # one frame per call: the coarse corrector sees the FIC ratio as the reference does
    """lock held for more than the 64 window searches the synchroniser remembers, then a dropout: k_acquire cannot replay sLevel
    from the last acquisition and brackets it instead (runs from 0 and from 3e38 over the remembered 64 frames); the two runs must
    have met, and the re-acquisition must land where the reference's does"""
    T_F = 196608
    nf = n_locked + 10
    x = synth.make_stream(nf, snr_db=20, cfo_hz=-45, delay=150, seed=21, subchs=synth.default_subchannels(2)).copy()
    x[n_locked * T_F + 30000:(n_locked + 1) * T_F + 90000] = 0
    o = R.orc_receiver_run(x)
    assert o["n_sync_false"] > 3
    d = d_factory(n_ensembles=1, max_frames=F, want_constellation=False)
    try:
        d.stream_upload(x[None, :])
        got = []
        for _ in range((nf + 8) // F + 4):
            d.process(F)
            info = d.frame_info()
            got += [(int(i["pos"]), int(i["start_index"]), int(i["fine"]), int(i["coarse"])) for i in info[0] if i["valid"] == 1]
        lost, ex = d.sync_stats()
        assert lost[0] >= 1 and d.relock_inexact[0] == 0, (lost, d.relock_inexact)
    finally:
        d.close()
    want = [(int(o["frame_pos"][k]), int(o["start_index"][k]), int(o["corr"][k][0]), int(o["corr"][k][1])) for k in range(o["n_frames"])]
    n = min(len(got), len(want))
    assert n >= o["n_frames"] - 1 and got[:n] == want[:n]


def check_dropout_relock(d_factory, F=1):
    """the signal disappears (samples squelched to zero) for 1.4 frames: the window search fails, the receiver searches for a null
    symbol with the sLevel the reference has at that moment -- it was advanced by every sample pulled while tracking -- lands on
    data symbols, fails again and re-locks; every attempt's position, window index, correctors and FIBs equal the oracle's"""
    T_F = 196608
    x = synth.make_stream(16, snr_db=18, cfo_hz=60, delay=200, seed=8).copy()
    x[6 * T_F + 50000:7 * T_F + 120000] = 0
    o = R.orc_receiver_run(x)
    assert o["n_sync_false"] > 5 and o["n_frames"] >= 13
    d = d_factory(n_ensembles=1, max_frames=F, want_constellation=False)
    try:
        d.stream_upload(x[None, :])
        got = []
        for _ in range(40 // F):
            d.process(F)
            info = d.frame_info(); fb, ok = d.fibs()
            for f in range(F):
                if info[0, f]["valid"] == 1:
                    got.append((int(info[0, f]["pos"]), int(info[0, f]["start_index"]), int(info[0, f]["fine"]), int(info[0, f]["coarse"]), ok[0, f].copy(), fb[0, f].copy()))
        lost, ex = d.sync_stats()
        assert lost[0] >= 2 and d.relock_inexact[0] == 0
    finally:
        d.close()
    n = len(got)
    assert n >= o["n_frames"] - 1
    ofib = o["fib"].reshape(-1, 12, 33)
    for k in range(min(n, o["n_frames"])):
        g = got[k]
        assert (g[0], g[1]) == (int(o["frame_pos"][k]), int(o["start_index"][k])), "frame %d found at %s, the reference finds it at %s" % (k, g[:2], (o["frame_pos"][k], o["start_index"][k]))
        assert (g[2], g[3]) == tuple(int(v) for v in o["corr"][k])
        assert np.array_equal(g[4], ofib[k, :, 0]) and np.array_equal(g[5], ofib[k, :, 1:])


def check_lock_lost_inside_a_replayed_batch(d_factory, F=8, pipeline_sync=0):
    """exact batch mode decodes a batch a second time, frame by frame, when a coarse-corrector decision was taken with a stale FIC ratio --
    and inside THAT batch the window search fails and the receiver re-acquires: the sLevel the null-symbol search starts with is replayed
    from the synchroniser's history ring (what every window search since the last acquisition pulled), which the first pass has already
    restarted over those entries -- the ring has to go back with the state.  Found by tools/sweep_independent.py (seed 2026, trial 33,
    ensemble 1: a random multiplex at 13 dB with -212.6 Hz offset, where the reference's coarse corrector runs wild from the ninth frame
    on; before the ring was put back the re-acquisition ended one sample early and the rest of the batch differed).  Every frame's
    position, window index, correctors and FIBs against the oracle."""
    u = [None, None, None, (4, 35, [(3, 24), (5, 17), (13, 12), (3, 17)]), None, (3, 29, [(3, 22), (4, 13), (14, 8), (3, 13)]), (28, 104, [(6, 24), (13, 18), (50, 13), (3, 19)]), None, None]
    spec = [(112, False, 3), (64, True, 3), (128, True, 1), (32, False, 1), (96, False, 2), (32, False, 2), (96, False, 1), (48, False, 3), (40, False, 3)]
    layout = []; cu = 0
    for i, (br, pb, lvl) in enumerate(spec):
        layout.append(synth.SubchannelCfg(i + 1, cu, br, pb, lvl, dabplus=False, uep=u[i])); cu += layout[-1].size_cu
    nf = 34
    x = synth.make_stream(nf, eid=0x5211, subchs=layout, snr_db=13.0, cfo_hz=-212.59620557244568, delay=71, seed=905702647)
    o = R.orc_receiver_run(x, subchs=[])
    assert o["n_sync_false"] >= 3 and o["n_frames"] >= 30, (o["n_sync_false"], o["n_frames"])
    d = d_factory(n_ensembles=1, max_frames=F, want_constellation=False, want_impulse_response=False, pipeline_sync=pipeline_sync)
    try:
        d.stream_upload(x[None, :])
        got = []; failed_in = []
        for step in range((nf - 2) // F):
            d.process(F)
            info = d.frame_info(); fb, ok = d.fibs()
            failed_in.append(bool((info["valid"] == 3).any()))
            for f in range(F):
                if info[0, f]["valid"] == 1:
                    got.append((int(info[0, f]["pos"]), int(info[0, f]["start_index"]), int(info[0, f]["fine"]), int(info[0, f]["coarse"]), ok[0, f].copy(), fb[0, f].copy()))
        assert d.replayed_batches() >= 1 and any(failed_in), (d.replayed_batches(), failed_in)     # (the batch of frames 24 .. 31 is the one that is both)
        lost, ex = d.sync_stats()
        assert lost[0] >= 2 and d.relock_inexact[0] == 0
    finally:
        d.close()
    ofib = o["fib"].reshape(-1, 12, 33)
    assert len(got) >= 28
    for k in range(min(len(got), o["n_frames"])):
        g = got[k]
        assert (g[0], g[1]) == (int(o["frame_pos"][k]), int(o["start_index"][k])), "frame %d found at %s, the reference finds it at %s" % (k, g[:2], (o["frame_pos"][k], o["start_index"][k]))
        assert (g[2], g[3]) == tuple(int(v) for v in o["corr"][k])
        assert np.array_equal(g[4], ofib[k, :, 0]) and np.array_equal(g[5], ofib[k, :, 1:])


def check_error_behaviour(d_factory):
    """signal problems never raise (they surface as valid = 0 / CRC false, like the reference's callbacks); programming errors
    come back as negative status codes with a message, never as a crash (the reference throws std::logic_error / out_of_range)"""
    from welle_io_amd.capi import DabPhyError
    # the documented limits of the batch geometry and of the configuration (include/dabphy.h): refused at creation, never wrapped silently
    for kw in (dict(n_ensembles=1, max_frames=4097), dict(n_ensembles=(1 << 22) // 64 + 1, max_frames=64), dict(n_ensembles=1, max_frames=1, decode_shape=4),
               dict(n_ensembles=0, max_frames=1), dict(n_ensembles=1, max_frames=1, pipeline_sync=4)):
        try:
            d_factory(**kw).close()
        except DabPhyError as e:
            assert "status -2" in str(e), (kw, str(e))
        else:
            raise AssertionError("dabphy_create accepted %s" % kw)
    d = d_factory(n_ensembles=2, max_frames=2)
    try:
        def bad(fn):
            try:
                fn()
            except DabPhyError as e:
                assert "status -" in str(e)
                return
            raise AssertionError("expected an error status")
        bad(lambda: d.process(1))                                                 # no stream bound
        bad(lambda: d.protection_eep(64, 0, 7))                                   # level out of range
        p = d.protection_eep(64, 0, 3)
        bad(lambda: d.set_subchannels([(1, 850, 48, p)]))                         # runs past CU 864
        bad(lambda: d.set_subchannels([(1, 0, 10, p)]))                           # too small for its protection profile
        x = np.zeros((2, 3 * 196608), np.complex64)
        d.stream_upload(x)
        bad(lambda: d.process(3))                                                 # more frames than max_frames
        d.process(2)                                                              # silence: no null symbol, no frame -- not an error
        assert (d.frame_info()["valid"] != 1).all()                             # (valid = 3: a window search that found nothing, reported like the reference does)
        bad(lambda: d.superframes(0, 64))                                         # no such sub-channel
        d.stream_open(4 * 196608)
        bad(lambda: d.stream_write(np.zeros((2, 5 * 196608), np.complex64)))      # more than the ring holds
        bad(lambda: d.stream_write_raw(np.zeros((2, 100, 2), np.uint8), "u8") if False else d._chk(d.lib.dabphy_stream_write_raw(d.h, None, 100, 1)))   # null buffer
    finally:
        d.close()


def check_timing_driver_refuses_a_stale_launch(d_factory):
    """dabphy_time_fused_msc re-runs the decode launch of the last batch; once a buffer that launch names has been replaced (new sub-channel
    classes, a reallocation) it must say so (DABPHY_ERR_STATE) instead of launching on freed memory"""
    from welle_io_amd.capi import DabPhyError
    subchs = synth.default_subchannels(3)
    x = synth.make_stream(5, subchs=subchs, snr_db=20, seed=4)
    d = d_factory(n_ensembles=1, max_frames=2)
    try:
        def refused():
            try:
                d.time_fused_msc(1)
            except DabPhyError as e:
                assert "status -5" in str(e), str(e)
                return True
            return False
        assert refused()                                                          # nothing decoded yet
        d.stream_upload(np.asarray(x, np.complex64)[None, :])
        d.set_subchannels([(s.subch_id, s.start_cu, s.size_cu, dev_prot(d, s)) for s in subchs])
        d.process(2)
        assert d.time_fused_msc(1) > 0.0
        d.set_subchannels([(s.subch_id, s.start_cu, s.size_cu, dev_prot(d, s)) for s in subchs[:2]])      # frees the classes' buffers
        assert refused()
        d.process(2)
        assert d.time_fused_msc(1) > 0.0
        assert d.time_copy(1 << 22, 0, 1) > 0.0                                   # the copy denominator's entry point (4 MiB: any device, the execution model too)
    finally:
        d.close()


# ---- the configuration bench.py times (welle_io_amd/workload.py): B x F batch, looping ring, coarse corrector enabled, pipelined
# synchroniser, all 18 sub-channels, superframe filter inside process() -- against the oracle on the very same samples
def check_bench_config(capi_mod, lib_path, B, F, pipeline_sync, check_ens, n_steps=3, demod_chunk=0, device="cuda", subs_idx=(0, 7, 17),
                       base=None, expect_chunk=None, channels=None, min_wide_fallbacks=None, decode_shape=0, min_chain_frames=None, cfo_max_hz=60.0, deferred_filter=False, sync_early=0):
    """deferred_filter: dabphy_set_auto_superframes(2) -- superframes_stats() returns the totals of the batch BEFORE the last process(); one more
    call at the end fetches the last batch's: the sums over the run are those of the immediate mode.
    channels: one channel (synth.apply_channel) per distinct recording -- the recordings then run through it ONCE over the whole test
    (a drifting sampling clock has no seamless loop point) and the ring does not loop.  min_wide_fallbacks: the wide synchroniser pass
    must have handed at least that many batches back to the frame-by-frame chain (what per-ensemble drift does in every batch)"""
    from welle_io_amd import workload
    loop = channels is None
    if channels is not None:
        nd = len(channels)
        base_np, txs = base if base is not None else workload.make_base_streams(nd, workload.REC_FRAMES, seed0=0)
        need = (n_steps * F + 8) * 196608
        rows = [synth.apply_channel(np.tile(base_np[e].astype(np.complex128), -(-(need + 4096) // base_np.shape[1])), channels[e])[:need] for e in range(nd)]
        base = (np.stack(rows).astype(np.complex64), txs)
    iq, cfo, base_np, txs = workload.make_batch(B, device=device, base=base, cfo_max_hz=cfo_max_hz)
    subchs = txs[0].subchs
    d = workload.open_receiver(capi_mod, lib_path, iq, F, subchs, pipeline_sync=pipeline_sync, demod_chunk=demod_chunk, profiling=False, loop=loop, decode_shape=decode_shape,
                               deferred_filter=deferred_filter, sync_early=sync_early)
    logs = {b: dict(fib=[], ok=[], corr=[], soft=[], msc=[[] for _ in subs_idx], sf=np.zeros(4, np.int64), n_logical=0) for b in check_ens}
    try:
        if expect_chunk is not None:
            assert d.demod_chunk() == expect_chunk, d.demod_chunk()
        for step in range(n_steps):
            d.process(F)
            info = d.frame_info(); fb, ok = d.fibs(); sf = d.superframes_stats()
            mscs = [d.msc(i) for i in subs_idx]
            for b in check_ens:
                L = logs[b]
                valid = [f for f in range(F) if info[b, f]["valid"] == 1]
                for f in valid:
                    L["fib"].append(fb[b, f]); L["ok"].append(ok[b, f]); L["corr"].append((int(info[b, f]["fine"]), int(info[b, f]["coarse"])))
                if valid:
                    f = valid[(step * 7) % len(valid)]          # all 230 400 soft bits of one frame per step, a different slot every step
                    L["soft"].append((len(L["fib"]) - len(valid) + valid.index(f), d.soft_bits(b, f)))
                for k in range(len(subs_idx)):
                    m, fv = mscs[k]
                    L["msc"][k].append(m[b, fv[b]:4 * len(valid)].tobytes())
                L["n_logical"] += max(0, 4 * len(valid) - int(mscs[0][1][b]))
                L["sf"] += sf[b]
        if deferred_filter:
            assert step == n_steps - 1
            sf = d.superframes_stats()                  # (the last batch's pass runs now)
            for b in check_ens:
                logs[b]["sf"] += sf[b]
            assert not d.superframes_stats().any()      # nothing pending, nothing unfetched: zeros
        wide = d.wide_sync_stats(); chain = d.find_chain_stats()
    finally:
        d.close()
    if min_wide_fallbacks is not None:
        assert wide[2] >= min_wide_fallbacks, "wide synchroniser pass: %d passes, %d handed back to the serial chain" % (wide[1], wide[2])
    if min_chain_frames is not None:
        # drifting windows: the searches ran in the find chain (k_sync_find_chain), the serial chain only where that could not help
        assert int(chain.sum()) >= min_chain_frames and (chain <= wide[0]).all(), "find chain: %s frames of %s accepted from the wide pass (%d passes, %d handed back to the serial chain)" % (chain, wide[0], wide[1], wide[2])
    loops = ((n_steps * F + 3) // (iq.shape[1] // 196608) + 2) if loop else 1
    for b in check_ens:
        L = logs[b]
        row = iq[b].cpu().numpy()
        o = R.orc_receiver_run(np.tile(row, loops), subchs=subchs, want_soft=True)
        n = len(L["fib"])
        assert n >= n_steps * F - (2 if loop else F + 2) and n <= o["n_frames"], (b, n, o["n_frames"])
        ofib = o["fib"][:12 * n].reshape(n, 12, 33)
        assert np.array_equal(np.array(L["ok"]), ofib[:, :, 0]), "ensemble %d: CRC flags differ" % b
        assert np.array_equal(np.array(L["fib"]), ofib[:, :, 1:]), "ensemble %d: FIB bytes differ" % b
        assert L["corr"] == [tuple(int(v) for v in c) for c in o["corr"][:n]], "ensemble %d: correctors differ" % b
        for k, soft in L["soft"]:
            assert np.array_equal(soft, o["soft"][k]), "ensemble %d frame %d: %d soft bits differ" % (b, k, (soft != o["soft"][k]).sum())
        for k, i in enumerate(subs_idx):
            got = b"".join(L["msc"][k])
            assert len(got) == L["n_logical"] * subchs[i].frame_bytes and len(got) > 0
            assert got == o["msc"][i][:len(got)], "ensemble %d: MSC bytes of sub-channel %d differ" % (b, i)
        # superframe filter totals over the same logical frames: every sub-channel through the restated SuperframeFilter
        want = np.zeros(4, np.int64)
        for i, sc in enumerate(subchs):
            fr = np.frombuffer(o["msc"][i], np.uint8)[:L["n_logical"] * sc.frame_bytes].reshape(-1, sc.frame_bytes)
            ev, _ = R.orc_superframe_run(fr)
            for e in ev:
                want += (e[3], e[1], e[2], (e[5] - bin(e[7]).count("1")) if e[3] else 0)
        assert tuple(L["sf"]) == tuple(want), "ensemble %d: superframe totals %s, oracle %s" % (b, tuple(L["sf"]), tuple(want))
    return logs


def check_mixed_layouts(capi_mod, lib_path, B, F, check_ens, n_steps=3, pipeline_sync=1, device="cuda", decode_shape=0, rec_frames=None):
    """A batch of INDEPENDENT ensembles (what every receiver of the reference is: its own MscHandler, msc-handler.cpp:61-127): ensemble b
    receives multiplex b % 5 of workload.mixed_layouts -- canonical, heterogeneous, two random ones, canonical again -- and selects ITS
    sub-channels (all; all; all; every other one; none) through dabphy_set_subchannels_ensemble.  FIBs, CRC flags, correctors, every
    selected sub-channel's bytes and the superframe totals of the ensembles in check_ens against the oracle on the very same samples."""
    from welle_io_amd import workload
    lib = capi_mod.load_library(lib_path)
    tx_lists, sel_lists = workload.mixed_layouts(lib)
    nd = len(tx_lists)
    base = workload.make_base_streams(nd, rec_frames or workload.rec_frames_for(F), seed0=70, subchs=tx_lists)
    iq, cfo, base_np, txs = workload.make_batch(B, device=device, base=base)
    sel = [sel_lists[b % nd] for b in range(B)]
    d = workload.open_receiver(capi_mod, lib_path, iq, F, sel, pipeline_sync=pipeline_sync, profiling=False, decode_shape=decode_shape)
    logs = {b: dict(fib=[], ok=[], corr=[], msc=[[] for _ in sel[b]], sf=np.zeros(4, np.int64), n_logical=0) for b in check_ens}
    pinned = None; pending = None
    try:
        for step in range(n_steps):
            d.process(F)
            info = d.frame_info(); fb, ok = d.fibs(); sf = d.superframes_stats()
            for b in check_ens:
                L = logs[b]
                valid = [f for f in range(F) if info[b, f]["valid"] == 1]
                for f in valid:
                    L["fib"].append(fb[b, f]); L["ok"].append(ok[b, f]); L["corr"].append((int(info[b, f]["fine"]), int(info[b, f]["coarse"])))
                for k in range(len(sel[b])):
                    m, fv, nr = d.msc_ensemble(b, k)
                    assert nr == 4 * len(valid)
                    L["msc"][k].append(m[fv:nr].tobytes())
                    if k == 0:
                        L["n_logical"] += max(0, nr - fv)      # (all of an ensemble's sub-channels were selected at the start of the stream: one first_valid)
                L["sf"] += sf[b]
            for b in range(B):
                if not sel[b]:
                    assert not sf[b].any(), "ensemble %d selects nothing but reports superframes %s" % (b, sf[b])
            # the bulk drain (dabphy_get_msc_batch: one copy per protection class + an index table, msc-handler.cpp:129-158 for a whole batch):
            # the same bytes and the same row windows as the per-service reads -- of EVERY service of the batch in the first step, of the
            # checked ensembles afterwards -- and the asynchronous form (dabphy_msc_drain_begin into page-locked memory, the NEXT
            # dabphy_process called while it is in flight, then dabphy_msc_drain_wait) delivers what the synchronous one did
            if pending is not None:
                d.msc_drain_wait()
                assert np.array_equal(pinned[:len(pending)], pending), "asynchronous bulk drain differs from the synchronous one (step %d)" % (step - 1)
            buf, desc = d.msc_batch()
            assert len(desc) == sum(len(l) for l in sel), (len(desc), sum(len(l) for l in sel))
            by = {(int(r["ensemble"]), int(r["subch_index"])): r for r in desc}
            assert len(by) == len(desc)
            for b in (range(B) if step == 0 else check_ens):
                for k, sc in enumerate(sel[b]):
                    m, fv, nr = d.msc_ensemble(b, k)
                    r = by[(b, k)]
                    assert (int(r["first_valid"]), int(r["n_rows"]), int(r["row_bytes"]), int(r["subch_id"])) == (fv, nr, sc.frame_bytes, sc.subch_id), (b, k, r, fv, nr)
                    rows = buf[int(r["offset"]):int(r["offset"]) + 4 * F * sc.frame_bytes].reshape(4 * F, sc.frame_bytes)
                    assert np.array_equal(rows[fv:nr], m[fv:nr]), "bulk drain: rows of ensemble %d sub-channel %d differ from dabphy_get_msc_ensemble's" % (b, k)
            if pinned is None:
                pinned = d.host_alloc((max(1, len(buf)),), np.uint8); pinned[:] = 0       # (the padding between two classes is never written)
            d.msc_drain_begin(pinned, np.zeros(len(desc), capi_mod.MSC_DESC_DTYPE))
            pending = buf.copy()
        d.msc_drain_wait()
        assert np.array_equal(pinned[:len(pending)], pending), "asynchronous bulk drain differs from the synchronous one (last step)"
    finally:
        if pinned is not None:
            d.msc_drain_wait(); d.host_free(pinned)
        d.close()
    loops = (n_steps * F + 3) // (iq.shape[1] // 196608) + 2
    for b in check_ens:
        L = logs[b]
        row = iq[b].cpu().numpy()
        o = R.orc_receiver_run(np.tile(row, loops), subchs=sel[b])
        n = len(L["fib"])
        assert n >= n_steps * F - 2 and n <= o["n_frames"], (b, n, o["n_frames"])
        ofib = o["fib"][:12 * n].reshape(n, 12, 33)
        assert np.array_equal(np.array(L["ok"]), ofib[:, :, 0]), "ensemble %d: CRC flags differ" % b
        assert np.array_equal(np.array(L["fib"]), ofib[:, :, 1:]), "ensemble %d: FIB bytes differ" % b
        assert L["corr"] == [tuple(int(v) for v in c) for c in o["corr"][:n]], "ensemble %d: correctors differ" % b
        want = np.zeros(4, np.int64)
        for k, sc in enumerate(sel[b]):
            got = b"".join(L["msc"][k])
            assert len(got) == L["n_logical"] * sc.frame_bytes and len(got) > 0, (b, k, len(got), L["n_logical"])
            assert got == o["msc"][k][:len(got)], "ensemble %d: MSC bytes of its sub-channel %d (%d kbit/s) differ" % (b, k, sc.bitrate)
            fr = np.frombuffer(o["msc"][k], np.uint8)[:L["n_logical"] * sc.frame_bytes].reshape(-1, sc.frame_bytes)
            ev, _ = R.orc_superframe_run(fr)
            for e in ev:
                want += (e[3], e[1], e[2], (e[5] - bin(e[7]).count("1")) if e[3] else 0)
        assert tuple(L["sf"]) == tuple(want), "ensemble %d: superframe totals %s, oracle %s" % (b, tuple(L["sf"]), tuple(want))
        assert not sel[b] or want[0] > 0, "ensemble %d: no superframe synchronised" % b
    return logs


def check_service_changes_in_mid_stream(d_factory, F=2, nf=26, snr_db=5.5, add_step=4, remove_step=8, pipeline_sync=False):
    """MscHandler::addSubchannel / removeSubchannel (msc-handler.cpp:61-127) while a batch of two receivers runs: ensemble 0 plays
    services A and B, adds D before step add_step and drops A before step remove_step; ensemble 1 plays C throughout.  The services
    that keep playing (B, C) must not notice: their bytes and their SuperframeFilter events (dabplus_decoder.cpp:50-213: a window that
    lives across batches) equal the uninterrupted oracle's, through both changes and the re-indexing they cause.  The new service D
    starts like a fresh DabAudio (dab-audio.cpp:146-149): first logical frame on the 17th CIF after the add, then the oracle's frames
    and the events of a SuperframeFilter that starts there."""
    xs, txs = [], []
    for e in range(2):
        x, tx = synth.make_stream(nf, eid=0x4000 + e, snr_db=snr_db, cfo_hz=(35, -60)[e], delay=(40, 700)[e], return_tx=True, seed=90 + e,
                                  payload_fn=synth.dabplus_payload_fn(80, 5 + e), noise_seed=78)   # (a noise sequence on which the reference's coarse corrector settles at this level)
        xs.append(x); txs.append(tx)
    n = min(len(x) for x in xs)
    xs = [x[:n] for x in xs]
    A, Bc, D = txs[0].subchs[2], txs[0].subchs[7], txs[0].subchs[12]
    Cc = txs[1].subchs[4]
    o0 = R.orc_receiver_run(xs[0], subchs=[A, Bc, D]); o1 = R.orc_receiver_run(xs[1], subchs=[Cc])
    want = {"A": o0["msc"][0], "B": o0["msc"][1], "D": o0["msc"][2], "C": o1["msc"][0]}
    d = d_factory(n_ensembles=2, max_frames=F, want_constellation=False, pipeline_sync=pipeline_sync)
    sub = lambda s: (s.subch_id, s.start_cu, s.size_cu, dev_prot(d, s))
    got = {k: dict(rows=[], cifs=[], ev=[], sf=[]) for k in "ABCD"}
    try:
        d.stream_upload(np.stack(xs))
        lists = [[("A", A), ("B", Bc)], [("C", Cc)]]
        d.set_subchannels_ensemble(0, [sub(s) for _, s in lists[0]]); d.set_subchannels_ensemble(1, [sub(s) for _, s in lists[1]])
        for step in range((nf + F - 1) // F):
            if step == add_step:
                lists[0] = [("A", A), ("B", Bc), ("D", D)]; d.set_subchannels_ensemble(0, [sub(s) for _, s in lists[0]])
            if step == remove_step:
                lists[0] = [("B", Bc), ("D", D)]; d.set_subchannels_ensemble(0, [sub(s) for _, s in lists[0]])
            d.process(F)
            info = d.frame_info()
            if not (info["valid"] == 1).any():
                break
            for b in range(2):
                c0 = 4 * int(info[b, 0]["frame_no"])
                for idx, (name, sc) in enumerate(lists[b]):
                    m, fv, nr = d.msc_ensemble(b, idx)
                    G = got[name]
                    base_row = len(G["rows"])
                    for r in range(fv, nr):
                        G["rows"].append(m[r].tobytes()); G["cifs"].append(c0 + r)
                    ev, ne, sf = d.superframes_ensemble(b, idx, sc.bitrate)
                    for k in range(ne):
                        e = ev[k]
                        # (event.cif = row of this batch; in terms of the service's own frame sequence: rows before first_valid do not count)
                        G["ev"].append((base_row + int(e["cif"]) - fv, int(e["corrected"]), int(e["uncorrectable"]), int(e["sync"]), int(e["format"]) if e["sync"] else 0, int(e["num_aus"]) if e["sync"] else 0,
                                        tuple(int(v) for v in e["au_start"][:e["num_aus"] + 1]) if e["sync"] else (), int(e["au_crc_ok"]) if e["sync"] else 0))
                        if e["sync"]:
                            G["sf"].append(sf[e["sf_slot"]].copy())
            if step == add_step:
                add_cif = 4 * int(info[0, 0]["frame_no"])
    finally:
        d.close()
    fb = 3 * Bc.bitrate

    def frames_of(buf, first=0):
        f = np.frombuffer(bytes(buf), np.uint8)
        return f[:len(f) // fb * fb].reshape(-1, fb)[first:]
    for name in "BCA":
        G = got[name]
        rows = b"".join(G["rows"])
        assert len(G["rows"]) >= (4 * (nf - 8) - 16 if name != "A" else 4 * F * (remove_step - 2) - 16), (name, len(G["rows"]))
        assert rows == bytes(want[name])[:len(rows)], "service %s: bytes differ from the uninterrupted reference stream" % name
        assert G["cifs"] == list(range(G["cifs"][0], G["cifs"][0] + len(G["cifs"]))) and G["cifs"][0] == 16, "service %s: a CIF is missing or repeated" % name
        eo, so = R.orc_superframe_run(frames_of(want[name])[:len(G["rows"])])
        assert G["ev"] == eo[:len(G["ev"])] and len(G["ev"]) >= len(eo) - 1 and len(eo) >= 3, "service %s: superframe events differ" % name
        assert all(np.array_equal(G["sf"][k], so[k]) for k in range(len(G["sf"]))) and len(G["sf"]) >= 2
    assert any(e[1] > 0 for e in got["B"]["ev"] + got["C"]["ev"]), "no byte error reached Reed-Solomon: raise the noise"
    G = got["D"]
    assert G["cifs"] and G["cifs"][0] == add_cif + 16, "the added service's first frame: CIF %s, expected %d" % (G["cifs"][:1], add_cif + 16)
    assert G["cifs"] == list(range(G["cifs"][0], G["cifs"][0] + len(G["cifs"])))
    fr = frames_of(want["D"], G["cifs"][0] - 16)
    rows = b"".join(G["rows"])
    assert len(G["rows"]) >= 4 * (nf - 4 - F * add_step) - 24 and rows == fr.tobytes()[:len(rows)], "added service: bytes differ"
    eo, so = R.orc_superframe_run(fr[:len(G["rows"])])
    assert G["ev"] == eo[:len(G["ev"])] and len(G["ev"]) >= len(eo) - 1 and len(eo) >= 2, "added service: superframe events differ"
    return got


def check_demod_chunks(d_factory, chunks=(1, 3, 7, 25, 38, 75), snr_db=13, seed=2, early=150):
    """explicit dabphy_config.demod_chunk values (work-groups of 7 / 25 / 75 data symbols; create() picks 25 for B x F >= 1024, the
    benchmark's case, and 15 otherwise): all 230 400 soft bits and the constellation taps equal the oracle's"""
    x = synth.make_stream(4, snr_db=snr_db, seed=seed)
    frames = cut_frames(x, 3, early)
    so, co, sn = R.orc_demod_frames(frames)
    for ch in chunks:
        d = d_factory(demod_chunk=ch)
        try:
            assert d.demod_chunk() == ch
            soft, con, snr = d.demod_frames(frames)
            assert np.array_equal(soft, so), "demod_chunk %d: %d soft bits differ" % (ch, (soft != so).sum())
            assert np.array_equal(con.view(np.uint32), co.view(np.uint32)), "demod_chunk %d: constellation points differ" % ch
        finally:
            d.close()


def check_ingest_vs_rawfile(d_factory, fmt, tmp_path, n_bytes=3 * 32768 + 4001, seed=77):
    """k_ingest against the reference's own CRAWFile reading the same bytes from a file: every byte value occurs, the byte count is
    odd (and not a multiple of the sample size), the file spans several of the reader's 32 KiB blocks, and the write wraps the ring"""
    rng = np.random.RandomState(seed)
    raw = rng.randint(0, 256, n_bytes).astype(np.uint8)
    raw[:512] = np.repeat(np.arange(256, dtype=np.uint8), 2)           # I = Q = every value
    raw[512:1024] = np.tile(np.array([0x80, 0x00, 0x7f, 0xff, 0x00, 0x80, 0xff, 0x7f], np.uint8), 64)   # extremes, both byte orders
    path = str(tmp_path / ("blob.%s.iq" % fmt))
    raw.tofile(path)
    bps = 2 if fmt in ("u8", "s8") else 4
    # without rewind the reference drops the file's last partial 32 KiB block (raw_file.cpp:307-325: a short read returns 0), so the
    # odd tail never reaches convertSamples: whole blocks are compared
    n = n_bytes // 32768 * 32768 // bps
    want = R.ref_rawfile_read(path, fmt, n)
    d = d_factory(n_ensembles=2, max_frames=1, want_constellation=False)
    try:
        ring = 4 * 196608
        d.stream_open(ring)
        pre = ring - n // 3                                              # fill up to n/3 samples before the wrap point
        d.stream_write_raw(np.zeros((2, pre, 2), np.uint8) + 128, "u8")
        body = raw[:n * bps].reshape(n, bps)
        both = np.stack([body, body[::-1]])                              # ensemble 1: the same samples in reverse order
        d.stream_write_raw(both, fmt)
        got0 = d.stream_read(0, pre, n); got1 = d.stream_read(1, pre, n)
    finally:
        d.close()
    assert np.array_equal(got0.view(np.uint32), want.view(np.uint32)), "%s: %d samples differ from CRAWFile::convertSamples" % (fmt, (got0 != want).sum())
    assert np.array_equal(got1.view(np.uint32), np.ascontiguousarray(want[::-1]).view(np.uint32))


def check_dropout_batch(d_factory, F=4):
    """the dropout of check_dropout_relock, four frames per call (an invalid slot in front of valid ones inside a batch).
    Positions, window indices and FIBs of every frame equal the oracle's as long as the coarse corrector's view of the FIC ratio agrees
    (it is disabled here: the stale-ratio deviation is pinned elsewhere); the logical frames of both sub-channels come out gap-free
    and equal the oracle's; the superframe filter's events equal the restated SuperframeFilter fed with those frames."""
    T_F = 196608
    x, tx = synth.make_stream(20, snr_db=18, cfo_hz=60, delay=200, seed=8, return_tx=True, payload_fn=synth.dabplus_payload_fn(80, 8))
    x = x.copy()
    x[6 * T_F + 50000:7 * T_F + 120000] = 0
    subs = [tx.subchs[1], tx.subchs[4]]
    o = R.orc_receiver_run(x, subchs=subs, disable_coarse=True)
    assert o["n_sync_false"] > 5 and o["n_frames"] >= 16
    d = d_factory(n_ensembles=1, max_frames=F, want_constellation=False, disable_coarse=True)
    try:
        d.stream_upload(x[None, :])
        d.set_subchannels([(s.subch_id, s.start_cu, s.size_cu, dev_prot(d, s)) for s in subs])
        got = []; msc = [[] for _ in subs]; sf_ev = [[] for _ in subs]; patterns = []
        for _ in range(12):
            d.process(F)
            info = d.frame_info(); fb, ok = d.fibs()
            if (info["valid"] == 0).all():
                break
            patterns.append(tuple(int(v) for v in info[0]["valid"]))
            for f in range(F):
                if info[0, f]["valid"] == 1:
                    got.append((int(info[0, f]["pos"]), int(info[0, f]["start_index"]), ok[0, f].copy(), fb[0, f].copy()))
            for i, sc in enumerate(subs):
                m, fv = d.msc(i)
                msc[i].append(m[0, fv[0]:d.msc_rows[0]].tobytes())
                ev, ne, sf = d.superframes(i, sc.bitrate)
                sf_ev[i] += [(int(e["corrected"]), int(e["uncorrectable"]), int(e["sync"])) for e in ev[0, :ne[0]]]
    finally:
        d.close()
    assert any(p.index(1) > 0 and p[0] != 1 for p in patterns if 1 in p) or any(3 in p and 1 in p[p.index(3):] for p in patterns), patterns   # an invalid slot in FRONT of valid ones
    n = len(got)
    assert n >= o["n_frames"] - F
    ofib = o["fib"].reshape(-1, 12, 33)
    for k in range(min(n, o["n_frames"])):
        g = got[k]
        assert (g[0], g[1]) == (int(o["frame_pos"][k]), int(o["start_index"][k])), "frame %d found at %s, the reference finds it at %s" % (k, g[:2], (o["frame_pos"][k], o["start_index"][k]))
        assert np.array_equal(g[2], ofib[k, :, 0]) and np.array_equal(g[3], ofib[k, :, 1:])
    for i, sc in enumerate(subs):
        b = b"".join(msc[i])
        assert len(b) == (4 * n - 16) * sc.frame_bytes and b == o["msc"][i][:len(b)], "MSC bytes of sub-channel %d differ" % i
        fr = np.frombuffer(b, np.uint8).reshape(-1, sc.frame_bytes)
        eo, _ = R.orc_superframe_run(fr)
        assert sf_ev[i] == [(e[1], e[2], e[3]) for e in eo], "superframe events of sub-channel %d differ" % i


def check_runtime_options(d_factory):
    """dabphy_set_options = OFDMProcessor::setReceiverOptions at run time (ofdm-processor.cpp:518-529): method changes apply from
    the next frame on without a restart; a change of disableCoarseCorrector restarts the synchroniser (correctors, phase, sLevel,
    acquisition) while the decoders keep their state; the stream goes on where the decoded frames ended"""
    x, tx = synth.make_stream(48, snr_db=18, cfo_hz=2060, delay=100, return_tx=True, seed=23, subchs=synth.default_subchannels(2))
    sent = [b"".join(f) for f in tx.fib_log]
    d = d_factory(n_ensembles=1, max_frames=3, want_constellation=False)
    try:
        d.stream_upload(x[None, :])
        d.set_subchannels([(s.subch_id, s.start_cu, s.size_cu, dev_prot(d, s)) for s in tx.subchs[:1]])
        frames = []

        def run(k):
            for _ in range(k):
                d.process(3)
                info = d.frame_info(); fb, ok = d.fibs()
                for f in range(3):
                    if info[0, f]["valid"] == 1:
                        frames.append((int(info[0, f]["frame_no"]), int(info[0, f]["coarse"]), bool(ok[0, f].all()), fb[0, f].tobytes()))
        run(3)
        assert d.config().fft_placement == 2 and d.config().disable_coarse == 0
        assert frames[-1][1] == 2000 and frames[-1][2]                    # the coarse corrector has pulled the 2.06 kHz offset in
        assert d.set_options(fft_placement=1, freqsync_method=1, disable_coarse=False) is False      # methods only: no restart
        assert d.config().fft_placement == 1 and d.config().freqsync_method == 1
        n0 = len(frames); run(2)
        assert len(frames) == n0 + 6 and all(f[2] for f in frames[n0:])   # goes on without losing a frame
        assert d.set_options(fft_placement=1, freqsync_method=1, disable_coarse=True) is True       # restart (resetCoarseCorrector + restart)
        assert d.config().disable_coarse == 1
        n1 = len(frames); run(4)
        after = frames[n1:]
        assert len(after) >= 8, len(after)
        assert all(f[1] == 0 for f in after)                             # coarse corrector reset to 0 and, disabled, never moved again
        assert [f[0] for f in frames] == list(range(len(frames)))        # the frame counter (CIF count of the de-interleavers) went on
        # without coarse correction the 2.06 kHz offset (two carrier spacings + 60 Hz) stays: FIBs no longer pass their CRC
        assert not any(f[2] for f in after[2:])
        assert d.set_options(fft_placement=2, freqsync_method=2, disable_coarse=False) is True
        n2 = len(frames); run(4)
        good = [f for f in frames[n2:] if f[2]]
        assert len(good) >= 6 and all(f[3] in sent for f in good)        # decodes the transmitted FIBs again
    finally:
        d.close()


def check_rs_random(d, n_sf=300, seed=77, s_per_sf=8):
    """RS(120,110): valid superframes with 0 .. 12 byte errors per code word at random positions (<= 5: corrected; more: the decoder
    gives up or miscorrects) -- corrected bytes, corrected-symbol totals and the uncorrectable verdict equal the oracle's, which is
    pinned to the reference's decode_rs_char (test_oracle_vs_ref.py::test_rs), word for word"""
    rng = np.random.RandomState(seed)
    sfs = np.stack([synth.make_superframe(8 * s_per_sf, rng, header=False) for _ in range(n_sf)])      # [n][120 * s]
    weights = []
    for k in range(n_sf):
        v = sfs[k].reshape(120, s_per_sf)
        for c in range(s_per_sf):
            w = int(rng.choice([0, 0, 1, 2, 3, 4, 5, 5, 6, 6, 7, 8, 10, 12]))
            pos = rng.choice(120, w, replace=False)
            v[pos, c] ^= rng.randint(1, 256, w).astype(np.uint8)
            weights.append(w)
    out, corr, unc = d.rs_superframes(sfs, s_per_sf)
    n_unc = 0
    for k in range(n_sf):
        o, c, u = R.orc_rs_superframe(sfs[k])
        assert np.array_equal(out[k], o), "superframe %d: corrected bytes differ" % k
        assert (int(corr[k]), bool(unc[k])) == (int(c), bool(u)), (k, int(corr[k]), int(unc[k]), c, u)
        n_unc += bool(u)
    assert n_unc > n_sf // 4 and max(weights) == 12          # the beyond-capacity paths were really taken


def check_wide_sync(d_factory, F=6, nf=34, pipeline_sync=False, cfo=37.0, snr_db=22, B=2):
    """The wide synchroniser pass (all frames of a batch at once from the predicted in-lock state, k_sync_validate) against the
    frame-by-frame chain and the oracle: same frames bit for bit; while the fine corrector still moves (the first ~20 frames at this
    offset, ofdm-processor.cpp:450-451) every pass hands over to the serial chain, once it rests whole batches are accepted."""
    logs_w, o, _ = check_stream_vs_oracle(d_factory, snr_db, cfo, 0, nf, False, B=B, F=F, pipeline_sync=pipeline_sync, seed=9)
    logs_s, _, _ = check_stream_vs_oracle(d_factory, snr_db, cfo, 0, nf, False, B=B, F=F, pipeline_sync=pipeline_sync, seed=9, serial_sync=True)
    # k_demod's two oscillator conversions (csrc/osc_exact.h) were both taken -- the soft bits above are the oracle's either way -- and
    # the checked one is the exception (a symbol in 25 or so reads a table entry next to a float rounding boundary)
    fast, checked = logs_w[0]["osc"]
    assert fast > 0 and checked > 0 and checked < 0.2 * fast, (fast, checked)
    for b in range(B):
        w, s = logs_w[b], logs_s[b]
        assert len(w["info"]) == len(s["info"])
        assert all(np.array_equal(np.array(w[k]), np.array(s[k])) for k in ("fib", "ok", "soft"))
        iw, isr = np.array(w["info"]), np.array(s["info"])
        assert all(np.array_equal(iw[k], isr[k], equal_nan=(k == "snr")) for k in iw.dtype.names)
        assert [bytes(m) for m in map(b"".join, w["msc"])] == [bytes(m) for m in map(b"".join, s["msc"])]
        wf, passes, fallbacks = w["wide"]
        assert s["wide"][0] == 0 and s["wide"][1] == 0                 # serial_sync: never queued
        # a frame is accepted from the wide pass iff no frame before it IN ITS BATCH moved a corrector: a batch that lies inside a run
        # of L frames with one fine-corrector value is accepted whole, and a run holds at least (L - 1 - (F - 1)) // F such batches
        fine = [int(v) for v in np.array(w["info"])["fine"]]
        runs = [len(list(g)) for _, g in __import__("itertools").groupby(fine)]
        whole = max(0, (max(runs) - F) // F)
        assert whole >= 1, fine                                        # (the stream is long enough to see an accepted batch)
        assert wf >= F * whole, (wf, fine)
        assert 1 <= fallbacks < passes, (passes, fallbacks)
    return logs_w


def check_exact_batch(d_factory, snr_db, cfo, F, seed, pipeline_sync=False, nf=25, B=1, expect_replay=None):
    """Exact batch mode (the default, dabphy_config.no_batch_replay = 0): batch mode with the reference's own FIC-ratio feedback.  The same low-SNR streams on which plain batch
    mode may part from the reference behind a stale coarse-corrector decision (check_stream_vs_oracle(ratio_lag_ok=True)) must now
    equal the oracle frame for frame, with NO tolerance: every FIB, corrector, soft bit, null symbol, MSC byte -- and the lag reports
    must come back empty, because a batch with an effective stale decision was put back and decoded again frame by frame."""
    logs, o, _ = check_stream_vs_oracle(d_factory, snr_db, cfo, 150, nf, False, B=B, F=F, seed=seed, pipeline_sync=pipeline_sync, exact_batch=True)
    for b in range(B):
        assert logs[b]["ratio_lag_effect"] == (0, -1), logs[b]["ratio_lag_effect"]
    if expect_replay is not None:
        assert (logs[0]["replayed"] > 0) == expect_replay, logs[0]["replayed"]
    return logs


def check_live_batch_vs_oracle(d_factory, F=3, nf=21, snr_db=3.5, cfo=-1000.0, seed=5, fmt="s16le", stats=None, ring_frames=None):
    """live ring (dabphy_stream_open / _write_raw, not a looping recording) decoded F frames per call: the synchroniser meets the end of
    the written samples inside a batch (slots without a frame), the wide pass has to leave those to nobody, and at this SNR exact batch
    mode decodes batches twice from the samples the first pass used.  FIBs and MSC bytes = the oracle's."""
    T_F = 196608
    x, tx = synth.make_stream(nf, snr_db=snr_db, cfo_hz=cfo, delay=300, return_tx=True, seed=seed)
    subs = [tx.subchs[2], tx.subchs[11]]
    raw, xf = raw_encode(x, fmt)
    o = R.orc_receiver_run(xf, subchs=subs)
    d = d_factory(n_ensembles=1, max_frames=F, want_constellation=False)
    try:
        # default: nothing leaves the ring, a re-acquisition can replay sLevel exactly; the feed still starves slots.
        # ring_frames: a ring shorter than the lock lasted -- the replay then brackets the level from [0, 2.125], the bound that holds for
        # samples converted from u8 / s8 / s16, and ~10 frames of history are enough for the two ends to meet
        ring = (ring_frames if ring_frames else nf + 4) * T_F
        d.stream_open(ring)
        d.set_subchannels([(s.subch_id, s.start_cu, s.size_cu, d.protection_eep(s.bitrate, s.profile_b, s.level)) for s in subs])
        fibs, oks, msc = [], [], [[] for _ in subs]
        wr = 0

        def feed(n):
            nonlocal wr
            n = min(n, len(raw) - wr)
            if n > 0:
                d.stream_write_raw(raw[wr:wr + n], fmt)
                wr += n
        feed((F + 1) * T_F + 1000)                         # a little more than a batch: the last slot of the first batches starves
        idle = 0
        while idle < 3:
            d.process(F)
            info = d.frame_info(); fb, ok = d.fibs()
            valid = [f for f in range(F) if info[0, f]["valid"] == 1]
            for f in valid:
                fibs.append(fb[0, f]); oks.append(ok[0, f])
            for i in range(len(subs)):
                m, fv = d.msc(i); msc[i].append(m[0, fv[0]:4 * len(valid)].tobytes())
            idle = 0 if (valid or wr < len(raw)) else idle + 1          # (out of lock the receiver may need several batches to find the next null symbol)
            room = ring - (wr - d.stream_consumed())
            feed(min(room - (T_F if ring_frames else 0), (F - 1) * T_F + 777))           # an uneven amount: batches with one, two or no starved slots
        n = len(fibs)
        assert n >= o["n_frames"] - 1, (n, o["n_frames"])
        ofib = o["fib"][:12 * n].reshape(n, 12, 33)
        assert np.array_equal(np.array(oks), ofib[:, :, 0]) and np.array_equal(np.array(fibs), ofib[:, :, 1:]), "FIBs differ"
        for i in range(len(subs)):
            got = b"".join(msc[i]); want = bytes(o["msc"][i])
            assert len(got) > 0 and got == want[:len(got)], "MSC bytes of sub-channel %d differ" % i
        if stats is not None:
            stats["replayed"] = d.replayed_batches(); stats["wide"] = d.wide_sync_stats()
            lost, _ = d.sync_stats(); stats["lost"] = int(lost[0]); stats["relock_inexact"] = int(d.relock_inexact[0])
    finally:
        d.close()


def check_exact_batch_mixed(d_factory, F=4, pipeline_sync=False, nf=25):
    """exact batch mode with DIFFERENT ensembles in one handle: ensemble 0 is the 3 dB stream whose second batch has to be decoded twice,
    ensemble 1 a clean 20 dB stream, ensemble 2 a 4 dB one -- the replay puts the state of ALL of them back and decodes the batch again;
    every ensemble must still equal ITS oracle run frame for frame (FIBs, CRC flags, correctors, MSC bytes)"""
    params = [(3, -1000.0, 5), (20, 37.0, 9), (4, 300.0, 3)]
    xs, orcs, subs = [], [], None
    for snr, cfo, seed in params:
        x, tx = synth.make_stream(nf, snr_db=snr, cfo_hz=cfo, delay=150, return_tx=True, seed=seed)
        subs = [tx.subchs[0], tx.subchs[9]]
        xs.append(x); orcs.append(R.orc_receiver_run(x, subchs=subs))
    n = max(len(x) for x in xs)
    xs = [np.concatenate([x, np.zeros(n - len(x), np.complex64)]) for x in xs]
    B = len(xs)
    d = d_factory(n_ensembles=B, max_frames=F, pipeline_sync=pipeline_sync, want_constellation=False)
    try:
        d.stream_upload(np.stack(xs))
        d.set_subchannels([(s.subch_id, s.start_cu, s.size_cu, dev_prot(d, s)) for s in subs])
        fib = [[] for _ in range(B)]; ok = [[] for _ in range(B)]; corr = [[] for _ in range(B)]; msc = [[[] for _ in subs] for _ in range(B)]
        for _ in range((nf + F - 1) // F + 3):
            d.process(F)
            info = d.frame_info(); fb, okk = d.fibs(); mscs = [d.msc(i) for i in range(len(subs))]
            for b in range(B):
                valid = [f for f in range(F) if info[b, f]["valid"] == 1]
                for f in valid:
                    fib[b].append(fb[b, f]); ok[b].append(okk[b, f]); corr[b].append((int(info[b, f]["fine"]), int(info[b, f]["coarse"])))
                for i in range(len(subs)):
                    m, fv = mscs[i]; msc[b][i].append(m[b, fv[b]:4 * len(valid)].tobytes())
        assert d.replayed_batches() >= 1
        eff, _ = d.ratio_lag_effect()
        assert (eff == 0).all(), eff
        for b in range(B):
            o = orcs[b]
            k = min(len(fib[b]), o["n_frames"] - 1)               # (the zero padding differs from the oracle's end of stream only after the last whole frame)
            assert k >= o["n_frames"] - 1 - F * (1 + {0: 0, 1: 1, 2: 1, 3: 2}[int(pipeline_sync)]), (b, k, o["n_frames"])
            ofib = o["fib"][:12 * k].reshape(k, 12, 33)
            assert np.array_equal(np.array(ok[b][:k]), ofib[:, :, 0]) and np.array_equal(np.array(fib[b][:k]), ofib[:, :, 1:]), "ensemble %d: FIBs differ" % b
            assert corr[b][:k] == [tuple(int(v) for v in c) for c in o["corr"][:k]], "ensemble %d: correctors differ" % b
            for i in range(len(subs)):
                got = b"".join(msc[b][i]); want = bytes(o["msc"][i])
                m = min(len(got), max(0, 4 * k - 16) * subs[i].frame_bytes)
                assert m > 0 and got[:m] == want[:m], "ensemble %d: MSC bytes of sub-channel %d differ" % (b, i)
    finally:
        d.close()


# ---------------------------------------------------------------------------------------------- channel impairments
# What the reference's own soak harness exercises (welle-cli/tests.cpp:305-370: multipath and the comparison of the FFT placement
# methods, phasereference.cpp:73-256) and what stresses the synchroniser's batch machinery here: the wide pass predicts "same window
# index, same correctors" for every frame of a batch, and a drifting sampling clock, a second path that outweighs the first, or a
# fading envelope break that prediction in most batches -- the frame-by-frame chain then takes over (redo_from), and the result must
# still be the oracle's, frame for frame, with NO tolerance.
CHANNELS = {
    "ppm+60":      dict(ppm=60.0),
    "ppm-100":     dict(ppm=-100.0),
    "echo300":     dict(echoes=[(300, 0.7)]),
    "pre-echo":    dict(echoes=[(-200, 0.6), (400, 0.4j)]),
    "sfn3":        dict(echoes=[(95, 0.9 * np.exp(0.7j)), (260, 0.55 * np.exp(-2.1j))]),
    "echo600":     dict(echoes=[(600, 0.8)]),                     # beyond the guard interval (504 samples)
    "fade7":       dict(fade=(0.3, 7.0)),
    "ppm+fade":    dict(ppm=40.0, fade=(0.3, 7.0), echoes=[(150, 0.5j)]),
}


def check_impaired_stream(d_factory, channel, F, schedule, placement, snr_db=18, cfo=55, delay=137, nf=None, seed=11, B=1, lockstep=False):
    """one stream through `channel`, decoded F frames per call in pipeline schedule `schedule` with FFT placement method `placement`,
    against the oracle: FIBs, CRC flags, correctors, constellation, null symbols, all soft bits, MSC bytes, SNR"""
    nf = nf or (3 * F + 4 if schedule == 0 else 4 * F + 5)
    check_stream_vs_oracle(d_factory, snr_db, cfo, delay, nf, lockstep, B=B, seed=seed, F=F, pipeline_sync=schedule, fft_placement=placement,
                           channel=CHANNELS[channel] if isinstance(channel, str) else channel)
