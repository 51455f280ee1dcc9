"""ctypes access to the checkers: oracle/_ref (the real reference, when prebuilt) and
oracle/libdabphy_oracle.so (our C restatement).  TEST INFRASTRUCTURE -- only tests/, smoke() and the
cpu_baseline leg of bench.py import this."""
import ctypes as C
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libwelle_ref.so")
REF_OFDM_SO = os.path.join(ROOT, "oracle", "_ref", "libwelle_ref_ofdm.so")


def have_ref():
    return os.path.exists(REF_SO) and os.path.exists(REF_OFDM_SO)


class RefSubch(C.Structure):
    _fields_ = [("subChId", C.c_int32), ("startAddr", C.c_int32), ("length", C.c_int32),
                ("shortForm", C.c_int32), ("uepTableIndex", C.c_int32), ("uepLevel", C.c_int32),
                ("eepProfileB", C.c_int32), ("eepLevel", C.c_int32), ("dabplus", C.c_int32),
                ("dump_path", C.c_char * 256)]


class RefRunIO(C.Structure):
    _fields_ = [("iq", C.c_void_p), ("n_samples", C.c_int64),
                ("disable_coarse", C.c_int32), ("fft_placement", C.c_int32), ("freqsync", C.c_int32),
                ("n_subch", C.c_int32), ("subch", C.POINTER(RefSubch)),
                ("fib", C.c_void_p), ("fib_cap", C.c_int32),
                ("cir", C.c_void_p), ("cir_cap", C.c_int32),
                ("con", C.c_void_p), ("con_cap", C.c_int32),
                ("nul", C.c_void_p), ("nul_cap", C.c_int32),
                ("snr", C.c_void_p), ("snr_cap", C.c_int32),
                ("corr", C.c_void_p), ("corr_cap", C.c_int32),
                ("n_fib", C.c_int32), ("n_cir", C.c_int32), ("n_con", C.c_int32), ("n_nul", C.c_int32),
                ("n_snr", C.c_int32), ("n_sync_true", C.c_int32), ("n_sync_false", C.c_int32),
                ("rs_calls", C.c_int32 * 16), ("rs_uncorr", C.c_int32 * 16), ("rs_corr", C.c_int32 * 16)]


_ref = None
_ref_ofdm = None


def ref():
    """the reference library; WELLE_REF_LIB selects a variant build (oracle/Makefile ref-variants: -O3, -DWITH_PROFILING)"""
    global _ref
    if _ref is None:
        _ref = C.CDLL(os.environ.get("WELLE_REF_LIB", REF_SO))
    return _ref


def ref_ofdm():
    global _ref_ofdm
    if _ref_ofdm is None:
        _ref_ofdm = C.CDLL(REF_OFDM_SO)
    return _ref_ofdm


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def level2_lib(variant, backend):
    """oracle/_ref/libwelle_l2{a,b}_{emu,hip}.so: the reference backend with ONE source replaced by a seam binding (INTEGRATION.md level 2)"""
    return os.path.join(ROOT, "oracle", "_ref", "libwelle_l2%s_%s.so" % (variant, backend))


def receiver_run(iq, subchs=(), dump_dir="/tmp", disable_coarse=False, fft_placement=2, freqsync=2, max_frames=None, lib=None):
    """Run the real reference RadioReceiver over a cf32 stream.  subchs: list of synth.SubchannelCfg.
    lib: another build of the same sources behind the same harness (level2_lib: one file replaced by a seam binding)"""
    iq = np.ascontiguousarray(iq, dtype=np.complex64)
    nf = max_frames or (len(iq) // 196608 + 2)
    io = RefRunIO()
    io.iq = _p(iq); io.n_samples = len(iq)
    io.disable_coarse = int(disable_coarse); io.fft_placement = fft_placement; io.freqsync = freqsync
    arr = (RefSubch * max(1, len(subchs)))()
    paths = []
    for i, s in enumerate(subchs):
        arr[i].subChId = s.subch_id; arr[i].startAddr = s.start_cu; arr[i].length = s.size_cu
        arr[i].shortForm = 0; arr[i].eepProfileB = int(s.profile_b); arr[i].eepLevel = s.level
        if getattr(s, "uep", None) is not None:
            arr[i].shortForm = 1; arr[i].uepTableIndex = s.uep[0]; arr[i].uepLevel = s.level
        arr[i].dabplus = int(s.dabplus)
        path = os.path.join(dump_dir, "refdump_%d_%d.msc" % (os.getpid(), s.subch_id))
        if os.path.exists(path):
            os.remove(path)
        arr[i].dump_path = path.encode()
        paths.append(path)
    io.n_subch = len(subchs); io.subch = arr
    fib = np.zeros((nf * 12, 33), np.uint8); io.fib = _p(fib); io.fib_cap = nf * 12
    cir = np.zeros((nf, 2048), np.float32); io.cir = _p(cir); io.cir_cap = nf
    con = np.zeros((nf, 1200), np.complex64); io.con = _p(con); io.con_cap = nf
    nul = np.zeros((nf, 2656), np.complex64); io.nul = _p(nul); io.nul_cap = nf
    snr = np.zeros(nf, np.float32); io.snr = _p(snr); io.snr_cap = nf
    corr = np.zeros((nf, 2), np.int32); io.corr = _p(corr); io.corr_cap = nf
    (C.CDLL(lib) if lib else ref()).ref_receiver_run(C.byref(io))
    msc = []
    for pth in paths:
        msc.append(open(pth, "rb").read() if os.path.exists(pth) else b"")
        if os.path.exists(pth):
            os.remove(pth)
    return dict(fib=fib[:io.n_fib], cir=cir[:io.n_cir], con=con[:io.n_con], nul=nul[:io.n_nul], snr=snr[:io.n_snr],
                corr=corr[:io.n_nul], n_sync_true=io.n_sync_true, n_sync_false=io.n_sync_false, msc=msc,
                rs_calls=list(io.rs_calls), rs_uncorr=list(io.rs_uncorr), rs_corr=list(io.rs_corr))


def ref_rawfile_read(path, fmt, n_samples):
    """the reference's CRAWFile (input/raw_file.cpp) reading `path` as `fmt`: the first n_samples it hands to getSamples()"""
    out = np.zeros(n_samples, np.complex64)
    n = ref().ref_rawfile_read(path.encode(), fmt.encode(), _p(out), n_samples)
    assert n == n_samples, (n, n_samples)
    return out


def ref_fft2048(x, inverse=False):
    a = np.ascontiguousarray(x, np.complex64).copy()
    ref().ref_fft2048(_p(a), int(inverse))
    return a


def ref_prs_reftable():
    a = np.zeros(2048, np.complex64); ref().ref_prs_reftable(_p(a)); return a


def ref_find_index(v, method=2):
    v = np.ascontiguousarray(v, np.complex64)
    ir = np.zeros(2048, np.float32)
    r = ref().ref_find_index(_p(v), method, _p(ir))
    return r, ir


def ref_freq_perm():
    a = np.zeros(1536, np.int16); ref().ref_freq_perm(_p(a)); return a


def ref_pcodes(idx):
    a = np.zeros(32, np.int8); ref().ref_pcodes(idx, _p(a)); return a


def ref_viterbi(soft, nbits):
    soft = np.ascontiguousarray(soft, np.int8); assert len(soft) == 4 * (nbits + 6)
    out = np.zeros(nbits, np.uint8); ref().ref_viterbi(_p(soft), nbits, _p(out)); return out


def ref_fic_decode(soft9216):
    soft = np.ascontiguousarray(soft9216, np.int8); assert soft.size == 9216
    bits = np.zeros((12, 256), np.uint8); ok = np.zeros(12, np.uint8)
    ratio = ref().ref_fic_decode(_p(soft), _p(bits), _p(ok))
    return bits, ok, ratio


def ref_eep(bitrate, profile_b, level, soft):
    soft = np.ascontiguousarray(soft, np.int8); out = np.zeros(24 * bitrate, np.uint8)
    ref().ref_eep_deconvolve(bitrate, int(profile_b), level, _p(soft), len(soft), _p(out)); return out


def ref_uep(bitrate, level, soft):
    soft = np.ascontiguousarray(soft, np.int8); out = np.zeros(24 * bitrate, np.uint8)
    ref().ref_uep_deconvolve(bitrate, level, _p(soft), len(soft), _p(out)); return out


def ref_energy(bits):
    a = np.ascontiguousarray(bits, np.uint8).copy(); ref().ref_energy_dedisperse(_p(a), len(a)); return a


def ref_rs_superframe(sf):
    a = np.ascontiguousarray(sf, np.uint8).copy(); c = C.c_int(0); u = C.c_int(0)
    ref().ref_rs_superframe(_p(a), len(a), C.byref(c), C.byref(u)); return a, c.value, u.value


def ref_ofdm_decode_frames(frames):
    """frames: (n, 2048+75*2552) cf32 -> soft (n,75,3072) int8, con (n,1200) cf64, snr list"""
    frames = np.ascontiguousarray(frames, np.complex64); n = frames.shape[0]
    soft = np.zeros((n, 75, 3072), np.int8); con = np.zeros((n, 1200), np.complex64); snr = np.zeros(n, np.float32)
    k = ref_ofdm().ref_ofdm_decode_frames(_p(frames), n, _p(soft), _p(con), _p(snr), n)
    return soft, con, snr[:k]


# ---------------------------------------------------------------------------------------------------
# our C restatement (oracle/libdabphy_oracle.so)
ORC_SO = os.path.join(ROOT, "oracle", "libdabphy_oracle.so")
_orc = None


class OrcProt(C.Structure):
    _fields_ = [("nbits", C.c_int), ("L", C.c_int * 4), ("PI", C.c_int * 4), ("n_in", C.c_int)]


class OrcSubchCfg(C.Structure):
    _fields_ = [("subch_id", C.c_int), ("start_cu", C.c_int), ("length_cu", C.c_int), ("prot", OrcProt)]


class OrcRunIO(C.Structure):
    _fields_ = [("iq", C.c_void_p), ("n_samples", C.c_int64), ("disable_coarse", C.c_int), ("fft_placement", C.c_int),
                ("n_subch", C.c_int), ("subch", C.POINTER(OrcSubchCfg)),
                ("fib", C.c_void_p), ("fib_cap", C.c_int), ("cir", C.c_void_p), ("cir_cap", C.c_int),
                ("con", C.c_void_p), ("con_cap", C.c_int), ("nul", C.c_void_p), ("nul_cap", C.c_int),
                ("snr", C.c_void_p), ("snr_cap", C.c_int), ("corr", C.c_void_p), ("corr_cap", C.c_int),
                ("start_index", C.c_void_p), ("frame_pos", C.c_void_p), ("sidx_cap", C.c_int),
                ("soft", C.c_void_p), ("soft_cap", C.c_int),
                ("msc", C.POINTER(C.c_void_p)), ("msc_cap", C.POINTER(C.c_int64)), ("msc_len", C.POINTER(C.c_int64)),
                ("n_fib", C.c_int), ("n_frames", C.c_int), ("n_snr", C.c_int), ("n_sync_true", C.c_int), ("n_sync_false", C.c_int), ("n_cir", C.c_int),
                ("freqsync_sel", C.c_int),
                ("tii_state", C.c_void_p), ("tii_rank", C.c_void_p), ("tii_ev", C.c_void_p), ("tii_cap", C.c_int), ("n_tii", C.c_int)]


TII_EVENT_DTYPE = np.dtype([("frame", "<i4"), ("comb", "<i4"), ("pattern", "<i4"), ("delay_samples", "<i4"), ("error", "<f4")])


def orc():
    global _orc
    if _orc is None:
        _orc = C.CDLL(ORC_SO)
        _orc.orc_init()
        for f in ("orc_twiddles_fwd", "orc_prs_reftable", "orc_freq_perm", "orc_pcodes", "orc_nco_table", "orc_prbs"):
            getattr(_orc, f).restype = C.c_void_p
    return _orc


def _arr(ptr, dtype, n):
    return np.frombuffer((C.c_char * (np.dtype(dtype).itemsize * n)).from_address(ptr), dtype=dtype).copy()


def orc_twiddles():
    return _arr(orc().orc_twiddles_fwd(), np.complex64, 2048)


def orc_prs_reftable():
    return _arr(orc().orc_prs_reftable(), np.complex64, 2048)


def orc_freq_perm():
    return _arr(orc().orc_freq_perm(), np.int16, 1536)


def orc_pcodes(idx):
    return _arr(orc().orc_pcodes(idx), np.int8, 32)


def orc_nco_table():
    return _arr(orc().orc_nco_table(), np.complex64, 2048000)


def orc_prbs(n):
    return _arr(orc().orc_prbs(n), np.uint8, n)


def orc_fft2048(x, inverse=False):
    a = np.ascontiguousarray(x, np.complex64); o = np.zeros(2048, np.complex64)
    orc().orc_fft2048(_p(a), _p(o), int(inverse)); return o


def orc_find_index(v, method=2):
    v = np.ascontiguousarray(v, np.complex64); ir = np.zeros(2048, np.float32)
    return orc().orc_find_index(_p(v), method, _p(ir)), ir


def orc_coarse_prs(v):
    v = np.ascontiguousarray(v, np.complex64); return orc().orc_coarse_prs(_p(v))


def orc_viterbi(soft, nbits):
    soft = np.ascontiguousarray(soft, np.int8); assert len(soft) == 4 * (nbits + 6)
    out = np.zeros(nbits, np.uint8); orc().orc_viterbi(_p(soft), nbits, _p(out)); return out


def orc_prot_fic():
    p = OrcProt(); orc().orc_prot_fic(C.byref(p)); return p


def orc_uep_row(bitrate, level):
    """(table index, size in CU) of the short-form table row for (bitrate, level)"""
    b = C.c_int(); l = C.c_int(); sz = C.c_int()
    for i in range(64):
        assert orc().orc_uep_table(i, C.byref(b), C.byref(l), C.byref(sz)) == 0
        if b.value == bitrate and l.value == level:
            return i, sz.value
    raise ValueError("no UEP row for %d kbit/s level %d" % (bitrate, level))


def orc_prot_of(s):
    """oracle protection record of a synth.SubchannelCfg (EEP long form or UEP short form)"""
    return orc_prot_uep(s.bitrate, s.level) if getattr(s, "uep", None) is not None else orc_prot_eep(s.bitrate, s.profile_b, s.level)


def uep_subchannel(synth, subch_id, start_cu, bitrate, level, dabplus=False):
    """a synth.SubchannelCfg in short form, its segments and size taken from the oracle's table (itself pinned to the reference's)"""
    p = orc_prot_uep(bitrate, level)
    idx, size = orc_uep_row(bitrate, level)
    segs = [(p.L[i], p.PI[i]) for i in range(4) if p.PI[i] > 0 and p.L[i] > 0]
    return synth.SubchannelCfg(subch_id, start_cu, bitrate, level=level, dabplus=dabplus, uep=(idx, size, segs))


def orc_prot_eep(bitrate, profile_b, level):
    p = OrcProt(); r = orc().orc_prot_eep(C.byref(p), bitrate, int(profile_b), level); assert r == 0; return p


def orc_prot_uep(bitrate, level):
    p = OrcProt(); orc().orc_prot_uep(C.byref(p), bitrate, level); return p


def orc_depuncture(prot, soft):
    soft = np.ascontiguousarray(soft, np.int8); out = np.zeros(4 * prot.nbits + 24, np.int8)
    orc().orc_depuncture(C.byref(prot), _p(soft), _p(out)); return out


def orc_msc_deconvolve(prot, soft):
    return orc_viterbi(orc_depuncture(prot, soft), prot.nbits)


def orc_fic_decode(soft9216, ratio=0):
    soft = np.ascontiguousarray(soft9216, np.int8); bits = np.zeros((12, 256), np.uint8); ok = np.zeros(12, np.uint8)
    r = C.c_int(ratio); orc().orc_fic_decode(_p(soft), _p(bits), _p(ok), C.byref(r)); return bits, ok, r.value * 10


def orc_rs_superframe(sf):
    a = np.ascontiguousarray(sf, np.uint8).copy(); c = C.c_int(0); u = C.c_int(0)
    orc().orc_rs_superframe(_p(a), len(a), C.byref(c), C.byref(u)); return a, c.value, u.value


def orc_rs_encode120(data110):
    d = np.ascontiguousarray(data110, np.uint8); par = np.zeros(10, np.uint8)
    orc().orc_rs_encode120(_p(d), _p(par)); return par


def orc_demod_frames(frames):
    """frames (n, 2048+75*2552) cf32 -> soft (n,75,3072), con (n,1200), snr list; state carried across frames"""
    frames = np.ascontiguousarray(frames, np.complex64); n = frames.shape[0]
    st = C.create_string_buffer(2048 * 8 + 16)
    orc().orc_demod_reset(st)
    soft = np.zeros((n, 75, 3072), np.int8); con = np.zeros((n, 75, 16), np.complex64); snrs = []
    for f in range(n):
        s = C.c_float(0)
        if orc().orc_demod_prs(st, _p(frames[f]), C.byref(s)):
            snrs.append(s.value)
        for k in range(75):
            sym = frames[f, 2048 + 2552 * k: 2048 + 2552 * (k + 1)]
            orc().orc_demod_symbol(st, _p(sym), _p(soft[f, k]), _p(con[f, k]))
    return soft, con.reshape(n, 1200), np.array(snrs, np.float32)


def orc_tii_rank():
    """[2][504] iteration rank of the reference's unordered_map<float, uint64_t> (oracle/tii_order.cpp)"""
    r = np.zeros((2, 504), np.int32); orc().orc_tii_iteration_rank(_p(r)); return r


def _tii_sorted(ev):
    return sorted((int(e["frame"]), int(e["comb"]), int(e["pattern"]), int(e["delay_samples"]), float(e["error"])) for e in ev)


def orc_tii_run(nulls, prss, want_detect=False):
    """TIIDecoder restatement over (NULL, PRS) pairs -> sorted events (frame, comb, pattern, delay_samples, error)"""
    nulls = np.ascontiguousarray(nulls, np.complex64); prss = np.ascontiguousarray(prss, np.complex64)
    lib = orc(); lib.orc_tii_state_bytes.restype = C.c_size_t
    st = np.zeros(lib.orc_tii_state_bytes(), np.uint8); rank = orc_tii_rank()
    out = []; det = np.zeros((len(nulls), 192), np.uint8)
    for i in range(len(nulls)):
        ev = np.zeros(16, TII_EVENT_DTYPE)
        n = lib.orc_tii_frame(_p(st), _p(nulls[i]), _p(prss[i]), _p(rank), _p(ev), 16, _p(det[i]))
        ev["frame"] = i
        out += list(ev[:n])
    out = _tii_sorted(out)
    return (out, det) if want_detect else out


def ref_tii_run(nulls, prss):
    """the reference's TIIDecoder class over the same pairs"""
    nulls = np.ascontiguousarray(nulls, np.complex64); prss = np.ascontiguousarray(prss, np.complex64)
    ev = np.zeros(16 * len(nulls) + 16, TII_EVENT_DTYPE)
    n = ref().ref_tii_run(_p(nulls), _p(prss), len(nulls), _p(ev), len(ev))
    assert n <= len(ev)
    return _tii_sorted(ev[:n])


def orc_receiver_run(iq, subchs=(), disable_coarse=False, fft_placement=2, want_soft=False, freqsync=2, tii=False):
    iq = np.ascontiguousarray(iq, np.complex64); nf = len(iq) // 196608 + 2
    io = OrcRunIO(); io.iq = _p(iq); io.n_samples = len(iq)
    io.disable_coarse = int(disable_coarse); io.fft_placement = fft_placement
    io.freqsync_sel = {2: 0, 0: 1, 1: 2}[freqsync]          # reference enum: 0 GetMiddle, 1 CorrelatePRS, 2 PatternOfZeros
    cfg = (OrcSubchCfg * max(1, len(subchs)))()
    bufs = []; ptrs = (C.c_void_p * max(1, len(subchs)))(); caps = (C.c_int64 * max(1, len(subchs)))(); lens = (C.c_int64 * max(1, len(subchs)))()
    for i, s in enumerate(subchs):
        cfg[i].subch_id = s.subch_id; cfg[i].start_cu = s.start_cu; cfg[i].length_cu = s.size_cu
        cfg[i].prot = orc_prot_of(s)
        b = np.zeros(nf * 4 * s.frame_bytes, np.uint8); bufs.append(b); ptrs[i] = b.ctypes.data; caps[i] = len(b)
    io.n_subch = len(subchs); io.subch = cfg; io.msc = ptrs; io.msc_cap = caps; io.msc_len = lens
    fib = np.zeros((nf * 12, 33), np.uint8); io.fib = _p(fib); io.fib_cap = nf * 12
    cir = np.zeros((nf, 2048), np.float32); io.cir = _p(cir); io.cir_cap = nf
    con = np.zeros((nf, 1200), np.complex64); io.con = _p(con); io.con_cap = nf
    nul = np.zeros((nf, 2656), np.complex64); io.nul = _p(nul); io.nul_cap = nf
    snr = np.zeros(nf, np.float32); io.snr = _p(snr); io.snr_cap = nf
    corr = np.zeros((nf, 2), np.int32); io.corr = _p(corr); io.corr_cap = nf
    sidx = np.zeros(nf, np.int32); fpos = np.zeros(nf, np.int64); io.start_index = _p(sidx); io.frame_pos = _p(fpos); io.sidx_cap = nf
    soft = np.zeros((nf if want_soft else 0, 75, 3072), np.int8)
    io.soft = _p(soft) if want_soft else None; io.soft_cap = nf if want_soft else 0
    if tii:
        orc().orc_tii_state_bytes.restype = C.c_size_t
        tst = np.zeros(orc().orc_tii_state_bytes(), np.uint8); trank = orc_tii_rank(); tev = np.zeros(16 * nf, TII_EVENT_DTYPE)
        io.tii_state = _p(tst); io.tii_rank = _p(trank); io.tii_ev = _p(tev); io.tii_cap = len(tev)
    orc().orc_receiver_run(C.byref(io))
    k = io.n_frames
    if tii:
        return dict(nul=nul[:k], tii=_tii_sorted(tev[:io.n_tii]), n_frames=k, fib=fib[:io.n_fib], corr=corr[:k])
    return dict(fib=fib[:io.n_fib], cir=cir[:min(io.n_cir, nf)], con=con[:k], nul=nul[:k], snr=snr[:io.n_snr], corr=corr[:k],
                start_index=sidx[:k], frame_pos=fpos[:k], soft=soft[:k] if want_soft else None,
                msc=[bufs[i][:lens[i]].tobytes() for i in range(len(subchs))],
                n_sync_true=io.n_sync_true, n_sync_false=io.n_sync_false, n_frames=k)


# ---------------------------------------------------------------------------------------------------
# our RadioReceiver drop-in (welle.io_amd/host) linked with the reference's FIBProcessor / DecoderAdapter objects
GPU_EMU_SO = os.path.join(ROOT, "oracle", "_ref", "libwelle_gpu_emu.so")
GPU_HIP_SO = os.path.join(ROOT, "oracle", "_ref", "libwelle_gpu_hip.so")


class GpuRunIO(C.Structure):
    _fields_ = [("iq", C.c_void_p), ("n_samples", C.c_int64), ("disable_coarse", C.c_int32), ("fft_placement", C.c_int32),
                ("n_subch", C.c_int32), ("subch", C.POINTER(RefSubch)),
                ("fib", C.c_void_p), ("fib_cap", C.c_int32), ("cir", C.c_void_p), ("cir_cap", C.c_int32),
                ("con", C.c_void_p), ("con_cap", C.c_int32), ("snr", C.c_void_p), ("snr_cap", C.c_int32),
                ("corr", C.c_void_p), ("corr_cap", C.c_int32),
                ("n_fib", C.c_int32), ("n_cir", C.c_int32), ("n_con", C.c_int32), ("n_snr", C.c_int32), ("n_corr", C.c_int32),
                ("n_sync_true", C.c_int32), ("n_sync_false", C.c_int32), ("n_services", C.c_int32),
                ("rs_calls", C.c_int32 * 16), ("rs_uncorr", C.c_int32 * 16), ("rs_corr", C.c_int32 * 16),
                ("nul", C.c_void_p), ("nul_cap", C.c_int32), ("n_nul", C.c_int32), ("freqsync", C.c_int32),
                ("decode_tii", C.c_int32), ("tii", C.c_void_p), ("tii_cap", C.c_int32), ("n_tii", C.c_int32)]


def gpu_receiver_run(iq, subchs=(), dump_dir="/tmp", disable_coarse=False, fft_placement=2, lib=GPU_EMU_SO, freqsync=2, tii=False):
    """Run GpuRadioReceiver (the facade mirror) over a cf32 stream; same outputs as receiver_run()."""
    L = C.CDLL(lib)
    iq = np.ascontiguousarray(iq, dtype=np.complex64)
    nf = len(iq) // 196608 + 2
    io = GpuRunIO(); io.iq = _p(iq); io.n_samples = len(iq); io.disable_coarse = int(disable_coarse); io.fft_placement = fft_placement; io.freqsync = freqsync
    arr = (RefSubch * max(1, len(subchs)))(); paths = []
    for i, s in enumerate(subchs):
        arr[i].subChId = s.subch_id; arr[i].startAddr = s.start_cu; arr[i].length = s.size_cu
        arr[i].shortForm = 0; arr[i].eepProfileB = int(s.profile_b); arr[i].eepLevel = s.level; arr[i].dabplus = int(s.dabplus)
        path = os.path.join(dump_dir, "gpudump_%d_%d.msc" % (os.getpid(), s.subch_id))
        if os.path.exists(path):
            os.remove(path)
        arr[i].dump_path = path.encode(); paths.append(path)
    io.n_subch = len(subchs); io.subch = arr
    fib = np.zeros((nf * 12, 33), np.uint8); io.fib = _p(fib); io.fib_cap = nf * 12
    cir = np.zeros((nf * 2, 2048), np.float32); io.cir = _p(cir); io.cir_cap = nf * 2
    con = np.zeros((nf, 1200), np.complex64); io.con = _p(con); io.con_cap = nf
    snr = np.zeros(nf, np.float32); io.snr = _p(snr); io.snr_cap = nf
    corr = np.zeros((nf, 2), np.int32); io.corr = _p(corr); io.corr_cap = nf
    nul = np.zeros((nf, 2656), np.complex64); io.nul = _p(nul); io.nul_cap = nf
    tev = np.zeros(16 * nf, TII_EVENT_DTYPE); io.decode_tii = int(tii); io.tii = _p(tev); io.tii_cap = len(tev)
    r = L.gpu_receiver_run(C.byref(io))
    assert r == 0, "GpuRadioReceiver run failed"
    msc = []
    for pth in paths:
        msc.append(open(pth, "rb").read() if os.path.exists(pth) else b"")
        if os.path.exists(pth):
            os.remove(pth)
    return dict(fib=fib[:io.n_fib], cir=cir[:io.n_cir], con=con[:io.n_con], snr=snr[:io.n_snr], corr=corr[:io.n_corr], msc=msc, nul=nul[:io.n_nul],
                n_sync_true=io.n_sync_true, n_sync_false=io.n_sync_false, n_services=io.n_services,
                rs_calls=list(io.rs_calls), rs_uncorr=list(io.rs_uncorr), rs_corr=list(io.rs_corr), tii=_tii_sorted(tev[:io.n_tii]))


# ---- DAB+ superframe filter (SuperframeFilter::Feed): oracle restatement and the real class
class SfEvent(C.Structure):
    _fields_ = [("cif", C.c_int32), ("corrected", C.c_int32), ("uncorrectable", C.c_int32), ("sync", C.c_int32), ("format", C.c_int32),
                ("num_aus", C.c_int32), ("au_start", C.c_int32 * 7), ("au_crc_ok", C.c_int32), ("sf_slot", C.c_int32)]


def _ev_tuple(e):
    return (e.cif, e.corrected, e.uncorrectable, e.sync, e.format if e.sync else 0, e.num_aus if e.sync else 0,
            tuple(e.au_start[:e.num_aus + 1]) if e.sync else (), e.au_crc_ok if e.sync else 0)


def orc_superframe_run(frames):
    """frames: [n][len] uint8 -> (list of event tuples, list of corrected superframes for the synced ones)"""
    frames = np.ascontiguousarray(frames, np.uint8); n, ln = frames.shape
    lib = orc()
    lib.orc_sf_state_bytes.restype = C.c_int
    st = np.zeros(lib.orc_sf_state_bytes(ln), np.uint8)
    out = np.zeros(5 * ln, np.uint8)
    evs, sfs = [], []
    for i in range(n):
        e = SfEvent()
        if lib.orc_superframe_feed(_p(st), _p(frames[i]), ln, i, C.byref(e), _p(out)):
            evs.append(_ev_tuple(e))
            if e.sync:
                sfs.append(out.copy())
    return evs, sfs


def ref_superframe_run(frames):
    frames = np.ascontiguousarray(frames, np.uint8); n, ln = frames.shape
    ev = (SfEvent * n)(); out = np.zeros((n, 5 * ln), np.uint8)
    ne = ref().ref_superframe_run(_p(frames), n, ln, ev, n, _p(out))
    return [_ev_tuple(ev[i]) for i in range(ne)], [out[i].copy() for i in range(ne) if ev[i].sync]


def ref_service_list_run(iq, realtime):
    """the real reference RadioReceiver over a stream, paced in real time or not -> (services listed at the end, onServiceDetected calls)"""
    iq = np.ascontiguousarray(iq, np.complex64)
    nl = C.c_int32(0); nd = C.c_int32(0)
    ref().ref_service_list_run.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    ref().ref_service_list_run(_p(iq), len(iq), int(realtime), C.byref(nl), C.byref(nd))
    return nl.value, nd.value


def ref_scan_run(iq):
    """onSignalPresence calls (1 / 0, in order) of the reference's RadioReceiver restarted in scan mode over a stream"""
    iq = np.ascontiguousarray(iq, np.complex64); calls = np.zeros(8, np.int32)
    ref().ref_scan_run.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
    n = ref().ref_scan_run(_p(iq), len(iq), _p(calls), 8)
    return [int(v) for v in calls[:min(n, 8)]]


def gpu_scan_run(iq, lib=GPU_EMU_SO):
    L = C.CDLL(lib)
    iq = np.ascontiguousarray(iq, np.complex64); calls = np.zeros(8, np.int32)
    L.gpu_scan_run.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
    n = L.gpu_scan_run(_p(iq), len(iq), _p(calls), 8)
    assert n >= 0
    return [int(v) for v in calls[:min(n, 8)]]


def gpu_failing_input_run(iq, fail_after, lib=GPU_EMU_SO):
    """GpuRadioReceiver over an input that throws after `fail_after` samples -> 1 if onInputFailure() was called and the receiver stopped cleanly"""
    L = C.CDLL(lib)
    iq = np.ascontiguousarray(iq, np.complex64)
    L.gpu_failing_input_run.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
    return L.gpu_failing_input_run(_p(iq), len(iq), int(fail_after))


def gpu_batch_services(iq, frames_per_step, n_steps, signal_clock=True, lib=GPU_EMU_SO):
    """GpuBatchReceiver with / without its signal-time clock -> per-ensemble (services listed, onServiceDetected calls)"""
    L = C.CDLL(lib)
    iq = np.ascontiguousarray(iq, np.complex64); B, n = iq.shape
    eid = np.zeros(B, np.int32); nl = np.zeros(B, np.int32); ok = np.zeros(B, np.int32); nd = np.zeros(B, np.int32); nt = np.zeros(B, np.int32)
    L.gpu_batch_run2.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    r = L.gpu_batch_run2(_p(iq), n, B, frames_per_step, n_steps, int(signal_clock), _p(eid), _p(nl), _p(ok), _p(nd), _p(nt))
    assert r == 0, "gpu_batch_run2 failed (%d)" % r
    return nl, nd


def gpu_node_run(iq, devices, frames_per_step, n_steps, lib=GPU_EMU_SO):
    """GpuNodeReceiver: [n_ens][n_samples] cf32 sharded by ensemble over one GpuBatchReceiver per entry of `devices`, the shards decoded
    concurrently -> per GLOBAL ensemble (eid, services listed, FIBs ok, onServiceDetected calls), number of shards"""
    L = C.CDLL(lib)
    iq = np.ascontiguousarray(iq, np.complex64); B, n = iq.shape
    dev = np.asarray(devices, np.int32)
    eid = np.zeros(B, np.int32); nl = np.zeros(B, np.int32); ok = np.zeros(B, np.int32); nd = np.zeros(B, np.int32); ns = np.zeros(1, np.int32)
    L.gpu_node_run.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    r = L.gpu_node_run(_p(iq), n, B, _p(dev), len(dev), frames_per_step, n_steps, _p(eid), _p(nl), _p(ok), _p(nd), _p(ns))
    assert r == 0, "gpu_node_run failed (%d)" % r
    return eid, nl, ok, nd, int(ns[0])


def gpu_batch_run(iq, frames_per_step, n_steps, lib=GPU_EMU_SO):
    """GpuBatchReceiver (one reference FIBProcessor per ensemble) over [n_ens][n_samples] cf32 -> per-ensemble (eid, services listed, FIBs ok, onServiceDetected calls)"""
    L = C.CDLL(lib)
    iq = np.ascontiguousarray(iq, np.complex64); B, n = iq.shape
    eid = np.zeros(B, np.int32); nl = np.zeros(B, np.int32); ok = np.zeros(B, np.int32); nd = np.zeros(B, np.int32); nt = np.zeros(B, np.int32)
    L.gpu_batch_run.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    r = L.gpu_batch_run(_p(iq), n, B, frames_per_step, n_steps, _p(eid), _p(nl), _p(ok), _p(nd), _p(nt))
    assert r == 0, "gpu_batch_run failed (%d)" % r
    return eid, nl, ok, nd, nt


class GpuBatchSub(C.Structure):
    _fields_ = [("ens", C.c_int32), ("add_step", C.c_int32), ("remove_step", C.c_int32), ("sub", RefSubch)]


def _fill_subch(r, s, path):
    r.subChId = s.subch_id; r.startAddr = s.start_cu; r.length = s.size_cu
    r.shortForm = 0; r.eepProfileB = int(s.profile_b); r.eepLevel = s.level
    if getattr(s, "uep", None) is not None:
        r.shortForm = 1; r.uepTableIndex = s.uep[0]; r.uepLevel = s.level
    r.dabplus = int(s.dabplus); r.dump_path = path.encode()


def gpu_batch_msc_run(iq, frames_per_step, n_steps, subs, lib=GPU_EMU_SO, dump_dir="/tmp"):
    """GpuBatchReceiver over [n_ens][n_samples] cf32 where every ensemble selects its own services: subs = [(ensemble, synth.SubchannelCfg,
    add_step, remove_step)] (add_step 0 = from the start, remove_step -1 = never) -> per entry (dump bytes of its DecoderAdapter,
    onRsErrors calls, uncorrectable ones, corrected symbols), FIBs ok per ensemble"""
    L = C.CDLL(lib)
    iq = np.ascontiguousarray(iq, np.complex64); B, n = iq.shape
    arr = (GpuBatchSub * max(1, len(subs)))(); paths = []
    for i, (e, s, a, r) in enumerate(subs):
        path = os.path.join(dump_dir, "gpubatch_%d_%d_%d.msc" % (os.getpid(), e, i))
        if os.path.exists(path):
            os.remove(path)
        arr[i].ens = e; arr[i].add_step = a; arr[i].remove_step = r
        _fill_subch(arr[i].sub, s, path); paths.append(path)
    calls = np.zeros(max(1, len(subs)), np.int32); unc = np.zeros_like(calls); corr = np.zeros_like(calls); ok = np.zeros(B, np.int32)
    L.gpu_batch_msc_run.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    r = L.gpu_batch_msc_run(_p(iq), n, B, frames_per_step, n_steps, arr, len(subs), _p(calls), _p(unc), _p(corr), _p(ok))
    assert r == 0, "gpu_batch_msc_run failed (%d)" % r
    out = []
    for i, pth in enumerate(paths):
        out.append((open(pth, "rb").read() if os.path.exists(pth) else b"", int(calls[i]), int(unc[i]), int(corr[i])))
        if os.path.exists(pth):
            os.remove(pth)
    return out, ok
