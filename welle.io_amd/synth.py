"""Synthetic DAB Mode-I transmit chain (the inverse of the receiver hot path).

The reference contains no modulator; this follows ETSI EN 300 401 as mirrored by the receiver code
(SURVEY.md Appendix B): FIB CRC (MathHelper.h:53-80), energy dispersal PRBS (fic-handler.cpp:62-71),
K=7 rate-1/4 mother code with the generator bit order of viterbi.cpp:36, puncturing vectors
(protTables.cpp:25-51; FIC fic-handler.cpp:158-191; EEP eep-protection.cpp:32-113), 16-CIF time
interleaving (dab-audio.cpp:113-143), CIF layout (msc-handler.cpp:129-158), QPSK mapping + frequency
interleaving (ofdm-decoder.cpp:198-214, freq-interleaver.cpp:35-59), differential modulation against
the phase reference symbol (phasereference.cpp:45-51) and OFDM symbol generation.

Used by tests/ and bench.py to make input; it is not part of the receive product path.
"""
import numpy as np

T_U, T_S, T_G, T_NULL, T_F, L_SYM, K_CARR = 2048, 2552, 504, 2656, 196608, 76, 1536
CIF_BITS = 55296
POLYS = (0o155, 0o117, 0o123, 0o155)
TI_MAP = np.array([0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15])


def prbs(n):
    """x^9 + x^5 + 1, all-ones start; bit i = s[8]^s[4] before the shift."""
    out = np.zeros(n, np.uint8)
    sr = [1] * 9
    for i in range(n):
        b = sr[8] ^ sr[4]
        sr = [b] + sr[:8]
        out[i] = b
    return out


_PRBS_CACHE = {}


def prbs_cached(n):
    if n not in _PRBS_CACHE:
        _PRBS_CACHE[n] = prbs(n)
    return _PRBS_CACHE[n]


def pi_vector(p):
    """Puncturing vector PI_p (p = 1..24), EN 300 401 table 29: 8+p ones in 32."""
    v = np.zeros(32, np.uint8)
    v[0::4] = 1
    order = [0, 4, 2, 6, 1, 5, 3, 7]
    for r in range(p):
        col = 1 + r // 8
        grp = order[r % 8]
        v[4 * grp + col] = 1
    return v


PI_X = np.array([1, 1, 0, 0] * 6, np.uint8)


def conv_encode(bits):
    """Mother code: returns 4*(n+6) bits, order (x0,x1,x2,x3) per input bit, 6 zero tail bits."""
    n = len(bits)
    padded = np.concatenate([np.zeros(6, np.uint8), bits.astype(np.uint8), np.zeros(6, np.uint8)])
    out = np.zeros((n + 6, 4), np.uint8)
    for k, poly in enumerate(POLYS):
        acc = np.zeros(n + 6, np.uint8)
        for j in range(7):
            if (poly >> j) & 1:
                acc ^= padded[6 - j: 6 - j + n + 6]
        out[:, k] = acc
    return out.reshape(-1)


def puncture_mask(segments, nbits):
    """segments: list of (n_blocks_of_128, PI index); returns bool mask of length 4*nbits+24."""
    parts = []
    for nblk, p in segments:
        if nblk > 0:
            parts.append(np.tile(pi_vector(p), 4 * nblk))
    m = np.concatenate(parts + [PI_X])
    assert len(m) == 4 * nbits + 24, (len(m), nbits)
    return m.astype(bool)


FIC_MASK = puncture_mask([(21, 16), (3, 15)], 768)


def eep_segments(bitrate, profile_b, level):
    """(L1,PI1),(L2,PI2) as eep-protection.cpp:32-113."""
    n = bitrate // 8
    if not profile_b:
        if level == 1:
            return [(6 * n - 3, 24), (3, 23)]
        if level == 2:
            if bitrate == 8:
                return [(5, 13), (1, 12)]
            return [(2 * n - 3, 14), (4 * n + 3, 13)]
        if level == 3:
            return [(6 * n - 3, 8), (3, 7)]
        if level == 4:
            return [(4 * n - 3, 3), (2 * n + 3, 2)]
    else:
        n32 = 24 * bitrate // 32 - 3
        pi = {4: (2, 1), 3: (4, 3), 2: (6, 5), 1: (10, 9)}[level]
        return [(n32, pi[0]), (3, pi[1])]
    raise ValueError("bad EEP level")


def crc16_ccitt(data):
    crc = 0xFFFF
    for b in data:
        crc ^= b << 8
        for _ in range(8):
            crc = ((crc << 1) ^ 0x1021) & 0xFFFF if crc & 0x8000 else (crc << 1) & 0xFFFF
    return crc ^ 0xFFFF


def make_fib(payload):
    """30 data bytes (padded with 0xFF) + inverted CRC-16-CCITT."""
    body = bytes(payload) + b"\xff" * (30 - len(payload))
    assert len(body) == 30
    c = crc16_ccitt(body)
    return body + bytes([c >> 8, c & 0xFF])


def bytes_to_bits(b):
    return np.unpackbits(np.frombuffer(bytes(b), np.uint8))


def fic_encode(fibs3):
    """3 FIBs (96 bytes) -> 2304 punctured bits."""
    bits = bytes_to_bits(b"".join(fibs3)) ^ prbs_cached(768)
    return conv_encode(bits)[FIC_MASK]


def msc_encode(frame_bytes, segments):
    """one logical frame (3*bitrate bytes) -> punctured bits (length*64)."""
    bits = bytes_to_bits(frame_bytes)
    n = len(bits)
    bits = bits ^ prbs_cached(n)
    return conv_encode(bits)[puncture_mask(segments, n)]


def freq_perm():
    t = np.zeros(T_U, np.int64)
    for i in range(1, T_U):
        t[i] = (13 * t[i - 1] + 511) % T_U
    keep = t[(t >= 256) & (t <= 256 + K_CARR) & (t != T_U // 2)]
    return keep - T_U // 2


_H = np.array([
    [0, 2, 0, 0, 0, 0, 1, 1, 2, 0, 0, 0, 2, 2, 1, 1] * 2,
    [0, 3, 2, 3, 0, 1, 3, 0, 2, 1, 2, 3, 2, 3, 3, 0] * 2,
    [0, 0, 0, 2, 0, 2, 1, 3, 2, 2, 0, 2, 2, 0, 1, 3] * 2,
    [0, 1, 2, 1, 0, 3, 3, 2, 2, 3, 2, 1, 2, 1, 3, 2] * 2])
# EN 300 401 table 39 (Mode I): for the 48 blocks of 32 carriers from k=-768 upward, (i, n)
_PRS_I = [0, 1, 2, 3] * 6 + [0, 3, 2, 1] * 6
_PRS_N = [1, 2, 0, 1, 3, 2, 2, 3, 2, 1, 2, 3, 1, 2, 3, 3, 2, 2, 2, 1, 1, 3, 1, 2,
          3, 1, 1, 1, 2, 2, 1, 0, 2, 2, 3, 3, 0, 2, 1, 3, 3, 3, 3, 0, 3, 0, 1, 1]


def prs_quadrant():
    """integer phase index q[bin] (phi = q*pi/2) for the 1536 active bins, -1 elsewhere."""
    q = -np.ones(T_U, np.int64)
    for blk in range(48):
        kmin = -768 + 32 * blk if blk < 24 else 1 + 32 * (blk - 24)
        for j in range(32):
            k = kmin + j
            q[k % T_U] = (_H[_PRS_I[blk]][j] + _PRS_N[blk]) % 4
    return q


def prs_freq():
    q = prs_quadrant()
    z = np.zeros(T_U, np.complex128)
    act = q >= 0
    z[act] = np.exp(1j * np.pi / 2 * q[act])
    return z


class SubchannelCfg:
    """EEP sub-channel (long form).  size in CUs derived like Subchannel::bitrate (dab-constants.cpp:404)."""

    def __init__(self, subch_id, start_cu, bitrate, profile_b=False, level=3, dabplus=True, uep=None):
        """uep = (table_index, size_cu, segments): short-form (UEP) sub-channel, `level` then is the UEP protection level and the
        (L_i, PI_i) segments / the size come from the caller (the tests take them from the oracle's table)"""
        self.subch_id, self.start_cu, self.bitrate = subch_id, start_cu, bitrate
        self.profile_b, self.level, self.dabplus = profile_b, level, dabplus
        self.uep = uep
        self.segments = eep_segments(bitrate, profile_b, level) if uep is None else list(uep[2])
        nb = 24 * bitrate
        self.n_coded = int(puncture_mask(self.segments, nb).sum())
        if uep is None:
            assert self.n_coded % 64 == 0
            self.size_cu = self.n_coded // 64
        else:
            self.size_cu = uep[1]                       # short form: the table's size; the punctured bits may fall short of it (padding)
            assert self.n_coded <= 64 * self.size_cu
        self.frame_bytes = 3 * bitrate

    def fig0_1(self):
        if self.uep is not None:                        # short form: table switch 0, 6-bit table index
            return bytes([(self.subch_id << 2) | (self.start_cu >> 8), self.start_cu & 0xFF, self.uep[0] & 0x3F])
        opt = 1 if self.profile_b else 0
        prot = self.level - 1
        return bytes([(self.subch_id << 2) | (self.start_cu >> 8), self.start_cu & 0xFF,
                      0x80 | (opt << 4) | (prot << 2) | (self.size_cu >> 8), self.size_cu & 0xFF])


def default_subchannels(n=18, bitrate=64):
    """canonical ensemble of SURVEY.md 8(d): n x 64 kbit/s DAB+ EEP-3A of 48 CU each."""
    out = []
    cu = 0
    for i in range(n):
        s = SubchannelCfg(i + 1, cu, bitrate)
        out.append(s)
        cu += s.size_cu
    assert cu <= 864
    return out


def build_fibs(eid, subchs, cif_count, extra_figs=()):
    """12 FIBs for one frame: FIG0/0, FIG0/1 (all sub-channels), FIG0/2, FIG1/0, FIG1/1, `extra_figs` (raw FIG bytes), rest padding."""
    figs = []
    figs.append(bytes([0x05, 0x00, eid >> 8, eid & 0xFF, (cif_count // 250) % 20, cif_count % 250]))
    for i in range(0, len(subchs), 6):
        body = b"".join(s.fig0_1() for s in subchs[i:i + 6])
        figs.append(bytes([len(body) + 1, 0x01]) + body)
    for i in range(0, len(subchs), 5):
        body = b""
        for s in subchs[i:i + 5]:
            sid = 0x1000 + s.subch_id
            body += bytes([sid >> 8, sid & 0xFF, 0x01, 0x3F if s.dabplus else 0x00, (s.subch_id << 2) | 0x02])
        figs.append(bytes([len(body) + 1, 0x02]) + body)
    label = ("MI355X ENS %04X" % eid).ljust(16)[:16].encode()
    figs.append(bytes([0x35, 0x00, eid >> 8, eid & 0xFF]) + label + b"\xff\x00")
    for s in subchs[:2]:
        sid = 0x1000 + s.subch_id
        figs.append(bytes([0x35, 0x01, sid >> 8, sid & 0xFF]) + ("SERVICE %02d" % s.subch_id).ljust(16).encode() + b"\xff\x00")
    figs += list(extra_figs)
    fibs = []
    for f in figs:          # first-fit packing
        for i in range(len(fibs)):
            if len(fibs[i]) + len(f) <= 30:
                fibs[i] += f
                break
        else:
            fibs.append(f)
    assert len(fibs) <= 12, len(fibs)
    while len(fibs) < 12:
        fibs.append(b"")
    return [make_fib(f) for f in fibs]


TII_PATTERNS = [v for v in range(256) if bin(v).count("1") == 4]     # the 70 octets of weight 4, ascending (table 67)


def tii_carriers(comb, pattern):
    """Carriers k (-768..768) of the TII signal of (comb, pattern) in Mode I, ascending: 4 blocks x 4 pairs."""
    ks = [1 + 2 * comb + 48 * b for b in range(8) if (TII_PATTERNS[pattern] >> (7 - b)) & 1]
    out = []
    for off in (-769, -385, 0, 384):
        for k in ks:
            out += [k + off, k + off + 1]
    return out


class EnsembleTx:
    """Generates consecutive Mode-I transmission frames (cf64 numpy arrays of T_F samples)."""

    def __init__(self, eid=0x1000, subchs=None, seed=0, payload_fn=None, amplitude=0.25, tii=None, tii_gain=1.0, extra_figs_fn=None):
        self.eid = eid
        self.extra_figs_fn = extra_figs_fn   # frame number -> list of extra FIGs (raw bytes) for that frame's FIC
        self.tii = tii                  # (comb 0..23, pattern 0..69): fills the null symbol (EN 300 401 clause 14.8)
        self.tii_gain = tii_gain
        self.subchs = default_subchannels() if subchs is None else subchs
        self.rng = np.random.RandomState(seed)
        self.payload_fn = payload_fn
        self.perm = freq_perm()
        self.bins = np.where(self.perm < 0, self.perm + T_U, self.perm)
        self.prs = prs_freq()
        self.cif_no = 0
        self.amplitude = amplitude
        # time interleaver memory: coded bits of the last 16 logical frames per sub-channel
        self.hist = {s.subch_id: np.zeros((16, s.n_coded), np.uint8) for s in self.subchs}
        self.fib_log = []       # list of 12x32-byte frames
        self.payload_log = {s.subch_id: [] for s in self.subchs}

    def _next_cif(self):
        cif = np.zeros(CIF_BITS, np.uint8)
        r = self.cif_no
        for s in self.subchs:
            if self.payload_fn is not None:
                data = self.payload_fn(s, r)
            else:
                data = self.rng.randint(0, 256, s.frame_bytes).astype(np.uint8).tobytes()
            self.payload_log[s.subch_id].append(bytes(data))
            coded = msc_encode(data, s.segments)
            h = self.hist[s.subch_id]
            h[r % 16] = coded
            # bit i of logical frame r' is sent in CIF r' + map[i%16]  ->  CIF r carries frame r - map[i%16]
            idx = np.arange(s.n_coded)
            src = (r - TI_MAP[idx % 16]) % 16
            tx = h[src, idx]
            tx[(r - TI_MAP[idx % 16]) < 0] = 0
            cif[s.start_cu * 64: s.start_cu * 64 + s.n_coded] = tx
        self.cif_no += 1
        return cif

    def next_frame_bits(self):
        extra = self.extra_figs_fn(len(self.fib_log)) if self.extra_figs_fn else ()
        fibs = build_fibs(self.eid, self.subchs, self.cif_no, extra)
        self.fib_log.append(fibs)
        fic = np.concatenate([fic_encode(fibs[3 * i:3 * i + 3]) for i in range(4)])
        msc = np.concatenate([self._next_cif() for _ in range(4)])
        return np.concatenate([fic, msc]).reshape(75, 3072)

    def next_frame(self):
        bits = self.next_frame_bits()
        z = self.prs.copy()
        syms = np.zeros((L_SYM, T_U), np.complex128)
        syms[0] = z
        for l in range(75):
            b = bits[l].astype(np.float64)
            y = ((1 - 2 * b[:K_CARR]) + 1j * (1 - 2 * b[K_CARR:])) / np.sqrt(2)
            z = z.copy()
            z[self.bins] = z[self.bins] * y
            syms[l + 1] = z
        t = np.fft.ifft(syms, axis=1) * T_U     # unnormalised inverse DFT
        t = np.concatenate([t[:, -T_G:], t], axis=1).reshape(-1)
        null = np.zeros(T_NULL, np.complex128)
        if self.tii is not None:
            zn = np.zeros(T_U, np.complex128)
            for k in tii_carriers(*self.tii)[0::2]:          # both carriers of a pair carry the PRS phase of the first
                zn[k % T_U] = zn[(k + 1) % T_U] = self.prs[k % T_U] * self.tii_gain
            tn = np.fft.ifft(zn) * T_U
            null = np.concatenate([tn[-(T_NULL - T_U):], tn])
        frame = np.concatenate([null, t])
        rms = np.sqrt(np.mean(np.abs(t) ** 2))
        return frame * (self.amplitude / rms)


def resample_ppm(x, ppm, taps=16):
    """x as a receiver whose sampling clock is `ppm` parts per million FAST sees it: y[n] = x(n * (1 + ppm * 1e-6)), band-limited
    interpolation (Hann-windowed sinc, `taps` taps).  The PRS of frame k then arrives k * 196608 * ppm * 1e-6 samples early."""
    if not ppm:
        return x
    n_out = int((len(x) - taps) / (1.0 + ppm * 1e-6))
    t = np.arange(n_out, dtype=np.float64) * (1.0 + ppm * 1e-6) + taps // 2
    k0 = np.floor(t).astype(np.int64)
    frac = t - k0
    y = np.zeros(n_out, np.complex128)
    for j in range(-taps // 2 + 1, taps // 2 + 1):
        u = frac - j                                               # distance from tap j to the interpolation point
        w = np.sinc(u) * (0.5 + 0.5 * np.cos(np.pi * u / (taps // 2)))
        y += x[k0 + j] * w
    return y


def apply_channel(x, channel):
    """The impairments the reference's own soak harness is built for (welle-cli/tests.cpp:305-370: multipath, FFT-window placement):
    channel = dict with any of
      echoes  [(delay_samples, complex gain), ...]  added to the direct path; a NEGATIVE delay is a pre-echo (the strongest path is
              not the first: what separates the three FFT placement methods, phasereference.cpp:73-256)
      fade    (depth, hz)  amplitude 1 + depth * cos(2 pi hz t): flat fading
      ppm     sampling-clock offset of the receiver (the window index then drifts from frame to frame)."""
    if not channel:
        return x
    y = x
    ech = channel.get("echoes")
    if ech:
        lead = max(0, -min(d for d, _ in ech))                    # pre-echoes: the whole signal moves back by the largest lead
        n = len(x) + lead
        y = np.zeros(n, np.complex128)
        y[lead:] += x
        for d, g in ech:
            o = lead + d
            m = min(len(x), n - o)
            y[o:o + m] += g * x[:m]
        y = y[:len(x)]
    fd = channel.get("fade")
    if fd:
        depth, hz = fd
        y = y * (1.0 + depth * np.cos(2 * np.pi * hz * np.arange(len(y)) / 2048000.0))
    if channel.get("ppm"):
        y = resample_ppm(y, channel["ppm"])
    return y


def make_stream(n_frames, eid=0x1000, subchs=None, seed=0, snr_db=None, cfo_hz=0.0, delay=0,
                noise_seed=1234, amplitude=0.25, payload_fn=None, return_tx=False, tii=None, extra_figs_fn=None, channel=None):
    """cf32 interleaved stream of n_frames frames (+ `delay` leading noise/zero samples).
    tii: None, or a list of transmitters (comb, pattern, delay_samples, gain) of a single-frequency network: identical
    frames, each with its own TII in the null symbol, summed with their relative delays.
    channel: multipath / fading / sampling-clock offset between transmitter and receiver (apply_channel)."""
    tx = EnsembleTx(eid, subchs, seed, payload_fn, amplitude, tii=tii[0][:2] if tii else None, extra_figs_fn=extra_figs_fn)
    x = np.concatenate([tx.next_frame() for _ in range(n_frames)])
    if tii:
        x = np.concatenate([np.zeros(tii[0][2], np.complex128), x])[:len(x)] * tii[0][3]
        for comb, pattern, d, g in tii[1:]:
            t2 = EnsembleTx(eid, subchs, seed, payload_fn, amplitude, tii=(comb, pattern))
            y = np.concatenate([t2.next_frame() for _ in range(n_frames)])
            x = x + g * np.concatenate([np.zeros(d, np.complex128), y])[:len(x)]
    x = apply_channel(x, channel)
    if delay:
        x = np.concatenate([np.zeros(delay, np.complex128), x])
    if cfo_hz:
        n = np.arange(len(x))
        x = x * np.exp(2j * np.pi * cfo_hz * n / 2048000.0)
    if snr_db is not None:
        rng = np.random.RandomState(noise_seed)
        sig_p = amplitude ** 2
        sigma = np.sqrt(sig_p / (10 ** (snr_db / 10)) / 2)
        x = x + sigma * (rng.randn(len(x)) + 1j * rng.randn(len(x)))
    out = x.astype(np.complex64)
    return (out, tx) if return_tx else out


def to_u8(x):
    """RAW u8 IQ as read by raw_file.cpp:324-366: (b-128)/128."""
    v = np.empty(2 * len(x), np.float32)
    v[0::2], v[1::2] = x.real, x.imag
    return np.clip(np.round(v * 127.0) + 128, 0, 255).astype(np.uint8)


# ---- DAB+ superframe payload with Reed-Solomon parity (EN 102 563 clause 6: RS(120,110) shortened from (255,245),
# ---- GF(2^8) polynomial 0x11D, generator roots alpha^0..alpha^9) so that the receiver's RS stage sees valid codewords
def _gf_tables():
    exp = np.zeros(512, np.int64); log = np.zeros(256, np.int64)
    x = 1
    for i in range(255):
        exp[i] = x; log[x] = i
        x <<= 1
        if x & 0x100:
            x ^= 0x11D
    exp[255:510] = exp[:255]
    return exp, log


_GF_EXP, _GF_LOG = _gf_tables()


def _rs_genpoly():
    g = [1]
    for r in range(10):
        ng = [0] * (len(g) + 1)
        for i, c in enumerate(g):          # multiply by (x + alpha^r)
            ng[i] ^= c
            if c:
                ng[i + 1] ^= int(_GF_EXP[_GF_LOG[c] + r])
        g = ng
    return g                               # highest degree first: g[0] = 1


_RS_GEN = _rs_genpoly()


def _rs_feedback_table():
    """_RS_FB[fb][j] = fb x g[j + 1] in GF(256): what one step of the systematic encoder adds to the remainder"""
    t = np.zeros((256, 10), np.uint8)
    for fb in range(1, 256):
        lf = _GF_LOG[fb]
        for j in range(10):
            c = _RS_GEN[j + 1]
            if c:
                t[fb, j] = _GF_EXP[lf + _GF_LOG[c]]
    return t


_RS_FB = _rs_feedback_table()


def rs_parity(data110):
    """systematic RS(120,110): 10 parity bytes for 110 data bytes -- or, for an array [110][n], the parity [10][n] of its n columns
    (the codewords of a superframe in one pass: the encoder's steps are table look-ups over all columns at once)"""
    d = np.asarray(data110, np.uint8)
    cols = d.reshape(110, -1)
    rem = np.zeros((cols.shape[1], 10), np.uint8)
    for k in range(110):
        fb = cols[k] ^ rem[:, 0]
        rem = np.concatenate([rem[:, 1:], np.zeros((cols.shape[1], 1), np.uint8)], axis=1) ^ _RS_FB[fb]
    return rem[0] if d.ndim == 1 else rem.T.copy()


def crc16(data, initial_invert, final_invert, poly):
    """CalcCRC of the reference (tools.cpp:41-72): MSB-first CRC-16"""
    crc = 0xFFFF if initial_invert else 0
    for b in bytes(data):
        crc ^= b << 8
        for _ in range(8):
            crc = ((crc << 1) ^ poly) & 0xFFFF if crc & 0x8000 else (crc << 1) & 0xFFFF
    return crc ^ 0xFFFF if final_invert else crc


def make_superframe(bitrate, rng, header=True):
    """a DAB+ audio superframe (ETSI TS 102 563) of 120*s bytes, s = bitrate/8: Fire-code protected header announcing
    48 kHz + SBR (3 access units), access units of random bytes each closed by its CRC-16-CCITT, and the 10 parity bytes of
    every column-interleaved RS(120,110) codeword (bytes pos*s+i).  header=False: random data with valid RS parity only."""
    s = bitrate // 8
    data = rng.randint(0, 256, 110 * s).astype(np.uint8)
    if header:
        n = 110 * s
        a0 = 6
        a1 = a0 + (n - a0) // 3 + int(rng.randint(0, 5)); a2 = a1 + (n - a0) // 3 - int(rng.randint(0, 5))
        data[2] = 0x40 | 0x20 | 0x10                       # dac_rate = 48 kHz, SBR, stereo -> 3 AUs, au_start[0] = 6
        data[3] = a1 >> 4; data[4] = ((a1 & 0xF) << 4) | (a2 >> 8); data[5] = a2 & 0xFF
        for lo, hi in ((a0, a1), (a1, a2), (a2, n)):
            # every access unit opens with ID_END (111): an AAC raw_data_block without a channel element, which a decoder rejects at
            # once.  Random bytes with a valid CRC are otherwise sometimes HALF accepted by FAAD2, and the reference's adapter then
            # throws on its decoder thread (dabplus_decoder.cpp:460) and takes the process down -- a crash above the PHY that would
            # limit which ensembles the real reference can be run on
            data[lo] |= 0xE0
            c = crc16(data[lo:hi - 2], True, True, 0x1021)
            data[hi - 2] = c >> 8; data[hi - 1] = c & 0xFF
        c = crc16(data[2:11], False, False, 0x782F)        # Fire code over bytes 2..10
        data[0] = c >> 8; data[1] = c & 0xFF
    sf = np.zeros(120 * s, np.uint8)
    sf[:110 * s] = data
    sf[110 * s:] = rs_parity(data.reshape(110, s)).reshape(-1)      # codeword i = bytes pos * s + i
    return sf


def dabplus_payload_fn(period_cifs=80, seed=0):
    """payload_fn for EnsembleTx: RS-valid DAB+ superframes, periodic with period_cifs (a multiple of 5 and of 16) so a
    recording of period_cifs/4 frames can be looped without breaking the time interleaver or the superframes"""
    assert period_cifs % 5 == 0 and period_cifs % 16 == 0
    cache = {}

    def fn(s, r):
        r = r % period_cifs
        q = r // 5
        key = (s.subch_id, q)
        if key not in cache:
            cache[key] = make_superframe(s.bitrate, np.random.RandomState(seed * 1000003 + s.subch_id * 1009 + q))
        fb = s.frame_bytes
        return cache[key][(r % 5) * fb:(r % 5 + 1) * fb].tobytes()
    return fn
