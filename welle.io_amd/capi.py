"""ctypes binding of the C ABI in include/dabphy.h (libdabphy_hip.so).

This is harness plumbing for tests/ and bench.py -- the product boundary is the C ABI itself, bound from the
reference's C++ host code as shown in INTEGRATION.md.  There is no CPU fallback: without the HIP library
(or without a gfx950 device) construction raises.
"""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "libdabphy_hip.so")

T_U, T_S, FRAME_SYMS_LEN = 2048, 2552, 2048 + 75 * 2552
ABI_VERSION = 6            # DABPHY_ABI_VERSION of the include/dabphy.h these structures mirror


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_ensembles", C.c_uint32), ("max_frames", C.c_uint32), ("device", C.c_int32),
                ("fft_placement", C.c_int32), ("disable_coarse", C.c_int32), ("want_constellation", C.c_int32),
                ("want_impulse_response", C.c_int32), ("demod_chunk", C.c_int32), ("freqsync_method", C.c_int32), ("pipeline_sync", C.c_int32), ("serial_sync", C.c_int32), ("no_batch_replay", C.c_int32), ("decode_shape", C.c_int32), ("sync_early", C.c_int32)]


class Protection(C.Structure):
    _fields_ = [("nbits", C.c_int32), ("L", C.c_int32 * 4), ("PI", C.c_int32 * 4)]


class Subchannel(C.Structure):
    _fields_ = [("subch_id", C.c_int32), ("start_cu", C.c_int32), ("size_cu", C.c_int32), ("prot", Protection)]


TII_DTYPE = np.dtype([("frame", "<i4"), ("comb", "<i4"), ("pattern", "<i4"), ("delay_samples", "<i4"), ("error", "<f4")])
SF_EVENT_DTYPE = np.dtype([("cif", "<i4"), ("corrected", "<i4"), ("uncorrectable", "<i4"), ("sync", "<i4"), ("format", "<i4"), ("num_aus", "<i4"),
                           ("au_start", "<i4", 7), ("au_crc_ok", "<i4"), ("sf_slot", "<i4")])


FRAME_INFO_RAW = np.dtype({"names": ["sample_pos", "frame_no", "start_index", "valid", "fine_corrector", "coarse_corrector", "snr"],
                           "formats": ["<i8", "<i8", "<i4", "<i4", "<i4", "<i4", "<f4"], "offsets": [0, 8, 16, 20, 24, 28, 32], "itemsize": 40})      # dabphy_frame_info
MSC_DESC_DTYPE = np.dtype([("ensemble", "<u4"), ("subch_index", "<u4"), ("row_bytes", "<u4"), ("first_valid", "<i4"), ("n_rows", "<i4"), ("subch_id", "<u4"), ("offset", "<u8")])


class DabPhyError(RuntimeError):
    pass


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def load_library(path=None):
    path = path or DEFAULT_LIB
    if not os.path.exists(path):
        raise DabPhyError("HIP library not built: %s (run __graft_entry__.build())" % path)
    lib = C.CDLL(path)
    lib.dabphy_last_error.restype = C.c_char_p
    lib.dabphy_device_name.restype = C.c_char_p
    return lib


class DabPhy:
    def __init__(self, n_ensembles=1, max_frames=1, device=0, lib_path=None, fft_placement=2, disable_coarse=False,
                 want_constellation=True, want_impulse_response=True, demod_chunk=0, pipeline_sync=False, freqsync_method=2, serial_sync=False, exact_batch=True, decode_shape=0, sync_early=0):
        self.lib = load_library(lib_path)
        cfg = Config(C.sizeof(Config), n_ensembles, max_frames, device, fft_placement, int(disable_coarse), int(want_constellation),
                     int(want_impulse_response), demod_chunk, freqsync_method, int(pipeline_sync), int(serial_sync), int(not exact_batch), int(decode_shape), int(sync_early))   # pipeline_sync: False/True/2
        self.cfg = cfg
        self.h = C.c_void_p()
        if self.lib.dabphy_abi_version() != ABI_VERSION:
            raise DabPhyError("library ABI version %d, this binding was written for %d" % (self.lib.dabphy_abi_version(), ABI_VERSION))
        r = self.lib.dabphy_create_v2(C.byref(cfg), C.byref(self.h))
        if r != 0:
            raise DabPhyError("dabphy_create failed with status %d (no gfx950 device?)" % r)

    def close(self):
        if self.h:
            self.lib.dabphy_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, r):
        if r != 0:
            raise DabPhyError("status %d: %s" % (r, self.lib.dabphy_last_error(self.h).decode()))

    @property
    def device_name(self):
        return self.lib.dabphy_device_name(self.h).decode()

    # ---- protection helpers
    def protection_fic(self):
        p = Protection(); self._chk(self.lib.dabphy_protection_fic(C.byref(p))); return p

    def protection_eep(self, bitrate, profile_b, level):
        p = Protection(); self._chk(self.lib.dabphy_protection_eep(C.byref(p), bitrate, int(profile_b), level)); return p

    def protection_uep(self, bitrate, level):
        p = Protection(); self._chk(self.lib.dabphy_protection_uep(C.byref(p), bitrate, level)); return p

    def protection_input_bits(self, p):
        return self.lib.dabphy_protection_input_bits(C.byref(p))

    # ---- unit-level seams
    def demod_frames(self, frames, want_con=True):
        frames = np.ascontiguousarray(frames, np.complex64).reshape(-1, FRAME_SYMS_LEN)
        n = frames.shape[0]
        soft = np.zeros((n, 75, 3072), np.int8)
        con = np.zeros((n, 1200), np.complex64) if want_con else None
        snr = np.zeros(n, np.float32)
        self._chk(self.lib.dabphy_demod_frames(self.h, _p(frames), n, _p(soft), _p(con) if want_con else None, _p(snr)))
        return soft, con, snr

    def viterbi_batch(self, soft, nbits):
        soft = np.ascontiguousarray(soft, np.int8).reshape(-1, 4 * (nbits + 6))
        n = soft.shape[0]
        out = np.zeros((n, nbits // 8), np.uint8)
        self._chk(self.lib.dabphy_viterbi_batch(self.h, _p(soft), nbits, n, _p(out)))
        return out

    def msc_deconvolve(self, prot, soft):
        nin = self.protection_input_bits(prot)
        soft = np.ascontiguousarray(soft, np.int8).reshape(-1, nin)
        n = soft.shape[0]
        out = np.zeros((n, prot.nbits // 8), np.uint8)
        self._chk(self.lib.dabphy_msc_deconvolve(self.h, C.byref(prot), _p(soft), n, _p(out)))
        return out

    def fic_decode(self, soft):
        soft = np.ascontiguousarray(soft, np.int8).reshape(-1, 9216)
        n = soft.shape[0]
        fib = np.zeros((n, 12, 32), np.uint8); ok = np.zeros((n, 12), np.uint8); ratio = C.c_int32(0)
        self._chk(self.lib.dabphy_fic_decode(self.h, _p(soft), n, _p(fib), _p(ok), C.byref(ratio)))
        return fib, ok, ratio.value

    # ---- diagnostics
    def time_demod(self, frames, n_ens, n_frames, mix=0, f_hz=0, iters=5):
        frames = np.ascontiguousarray(frames, np.complex64).reshape(-1, FRAME_SYMS_LEN)
        ms = C.c_float(0)
        self._chk(self.lib.dabphy_time_demod(self.h, _p(frames), frames.shape[0], n_ens, n_frames, mix, f_hz, iters, C.byref(ms)))
        return ms.value

    def superframes(self, subch_index, bitrate):
        """-> (events structured array [B][4F], n_events [B], corrected superframes [B][n_slots][120*bitrate/8])"""
        B, F = self.cfg.n_ensembles, self._last
        ev = np.zeros((B, 4 * F), SF_EVENT_DTYPE); ne = np.zeros(B, np.int32)
        sf = np.zeros((B, 4 * F // 5 + 1, 15 * bitrate), np.uint8)
        self._chk(self.lib.dabphy_superframes(self.h, subch_index, _p(ev), _p(ne), _p(sf)))
        return ev, ne, sf

    def sync_stats(self):
        """(failed window searches, frames settled by the ordered float sums) per ensemble since reset"""
        lost = np.zeros(self.cfg.n_ensembles, np.int32); ex = np.zeros(self.cfg.n_ensembles, np.int32)
        self.relock_inexact = np.zeros(self.cfg.n_ensembles, np.int32)
        self._chk(self.lib.dabphy_get_sync_stats(self.h, _p(lost), _p(ex), _p(self.relock_inexact)))
        return lost, ex

    def wide_sync_stats(self):
        """(frames per ensemble accepted from the wide synchroniser pass, passes queued, passes that needed the serial chain)"""
        wf = np.zeros(self.cfg.n_ensembles, np.int32); a = C.c_uint64(0); b = C.c_uint64(0)
        self._chk(self.lib.dabphy_get_wide_sync_stats(self.h, _p(wf), C.byref(a), C.byref(b)))
        return wf, a.value, b.value

    def find_chain_stats(self):
        """frames per ensemble whose window search ran in the find chain (of those accepted from the wide synchroniser pass)"""
        cf = np.zeros(self.cfg.n_ensembles, np.int32)
        self._chk(self.lib.dabphy_get_find_chain_stats(self.h, _p(cf)))
        return cf

    def replayed_batches(self):
        a = C.c_uint64(0)
        self._chk(self.lib.dabphy_get_replayed_batches(self.h, C.byref(a)))
        return a.value

    def wide_superframe_stats(self):
        """(sub-channel batches the superframe filter's wide pass settled, batches it was tried on) since create"""
        a = C.c_uint64(0); b = C.c_uint64(0)
        self._chk(self.lib.dabphy_get_wide_superframe_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def osc_stats(self):
        """(symbols mixed with the unchecked oscillator conversion, symbols that took the checked one) since create"""
        a = C.c_uint64(0); b = C.c_uint64(0)
        self._chk(self.lib.dabphy_get_osc_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_track_slevel(self, on=True):
        self._chk(self.lib.dabphy_set_track_slevel(self.h, int(on)))

    def set_auto_superframes(self, on=True):        # (on = 2: the filter pass of a batch deferred to the next process(), beside its FFT stage)
        """run the all-sub-channel superframe filter inside every process() call; superframes_stats() then only fetches the totals"""
        self._chk(self.lib.dabphy_set_auto_superframes(self.h, int(on)))

    def superframes_stats(self):
        st = np.zeros((self.cfg.n_ensembles, 4), np.int32)
        self._chk(self.lib.dabphy_superframes_stats(self.h, _p(st)))
        return st

    def set_tii(self, on=True):
        """RadioReceiverOptions::decodeTII for the following process() calls"""
        self._chk(self.lib.dabphy_set_tii(self.h, int(on)))

    def tii(self, max_per_ensemble=None):
        """onTIIMeasurement calls of the last batch: list per ensemble of records (frame, comb, pattern, delay_samples, error)"""
        B = self.cfg.n_ensembles
        m = 9 * self.cfg.max_frames if max_per_ensemble is None else max_per_ensemble
        ev = np.zeros((B, max(m, 1)), TII_DTYPE); n = np.zeros(B, np.int32)
        self._chk(self.lib.dabphy_get_tii(self.h, _p(ev), _p(n), m))
        return [ev[b, :min(int(n[b]), m)].copy() for b in range(B)], n

    def selftest_unit_twiddle(self):
        c = (C.c_uint64 * 2)()
        self._chk(self.lib.dabphy_selftest_unit_twiddle(self.h, c))
        return [int(x) for x in c]

    def selftest_pair_exchange(self):
        c = (C.c_uint64 * 2)()
        self._chk(self.lib.dabphy_selftest_pair_exchange(self.h, c))
        return [int(x) for x in c]

    def selftest_div127(self):
        c = (C.c_uint64 * 3)()
        self._chk(self.lib.dabphy_selftest_div127(self.h, c))
        return [int(x) for x in c]

    def time_viterbi(self, nbits, n_codewords, iters=3):
        a = C.c_float(0); b = C.c_float(0)
        self._chk(self.lib.dabphy_time_viterbi(self.h, nbits, n_codewords, iters, C.byref(a), C.byref(b)))
        return a.value, b.value

    def time_copy(self, nbytes=2 << 30, blocks_per_cu=0, iters=5):
        """GB/s (read + write) of a plain float4 device copy of nbytes (dabphy_time_copy)"""
        a = C.c_float(0)
        self._chk(self.lib.dabphy_time_copy(self.h, C.c_uint64(nbytes), blocks_per_cu, iters, C.byref(a)))
        return a.value

    def traceback_split(self, on=True):
        """the lane-per-code-word kernel's traceback as a pass of its own beside the forward pass (dabphy_test_traceback_split)"""
        self._chk(self.lib.dabphy_test_traceback_split(self.h, int(on)))

    def last_decode_plan(self):
        """(kernel that decoded the last process(), numbered like dabphy_config.decode_shape: 1 = lane per code word (k_viterbi_fused), 2 = state-parallel with two code words per wavefront (k_viterbi_sp2 + k_traceback_sp2), 3 = state-parallel with one (k_viterbi_sp), 0 = nothing decoded yet; protection classes in that launch)"""
        a = C.c_int32(0); b = C.c_int32(0)
        self._chk(self.lib.dabphy_last_decode_plan(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def time_fused_msc(self, iters=3):
        a = C.c_float(0)
        self._chk(self.lib.dabphy_time_fused_msc(self.h, iters, C.byref(a)))
        return a.value

    # ---- streaming receiver
    def reset(self):
        self._chk(self.lib.dabphy_reset(self.h))

    def stream_upload(self, iq, loop=False):
        """iq: (n_ensembles, n_samples) complex64 host array"""
        iq = np.ascontiguousarray(iq, np.complex64).reshape(self.cfg.n_ensembles, -1)
        self._chk(self.lib.dabphy_stream_upload(self.h, _p(iq), C.c_uint64(iq.shape[1]), int(loop)))

    def stream_bind_device(self, dev_ptr, ring_samples, stride_samples, n_valid, loop=False):
        self._chk(self.lib.dabphy_stream_bind_device(self.h, C.c_void_p(dev_ptr), C.c_uint64(ring_samples), C.c_uint64(stride_samples),
                                                     C.c_uint64(n_valid), int(loop)))

    def set_subchannels(self, subs):
        """subs: list of (subch_id, start_cu, size_cu, Protection)"""
        arr = (Subchannel * max(1, len(subs)))()
        for i, (sid, start, size, prot) in enumerate(subs):
            arr[i].subch_id = sid; arr[i].start_cu = start; arr[i].size_cu = size; arr[i].prot = prot
        self._n_sub = len(subs); self._sub_bytes = [s[3].nbits // 8 for s in subs]; self._ens_bytes = {}
        self._chk(self.lib.dabphy_set_subchannels(self.h, arr, len(subs)))

    def set_subchannels_ensemble(self, ensemble, subs):
        """the list of ONE ensemble of the batch (dabphy_set_subchannels_ensemble): takes effect with the next process()"""
        arr = (Subchannel * max(1, len(subs)))()
        for i, (sid, start, size, prot) in enumerate(subs):
            arr[i].subch_id = sid; arr[i].start_cu = start; arr[i].size_cu = size; arr[i].prot = prot
        self._ens_bytes = getattr(self, "_ens_bytes", {})
        self._ens_bytes[ensemble] = [s[3].nbits // 8 for s in subs]
        self._chk(self.lib.dabphy_set_subchannels_ensemble(self.h, ensemble, arr, len(subs)))

    def msc_ensemble(self, ensemble, idx):
        """-> (out [4F][bytes], first_valid, n_rows) of sub-channel idx (position in ITS list) of one ensemble"""
        F = self._last
        nb = self._ens_bytes[ensemble][idx] if ensemble in getattr(self, "_ens_bytes", {}) else self._sub_bytes[idx]
        out = np.zeros((4 * F, nb), np.uint8); fv = C.c_int32(0); nr = C.c_int32(0)
        self._chk(self.lib.dabphy_get_msc_ensemble(self.h, ensemble, idx, _p(out), C.c_size_t(out.nbytes), C.byref(fv), C.byref(nr)))
        return out, fv.value, nr.value

    def msc_batch_size(self):
        """(bytes, services) the bulk drain of the last batch needs (dabphy_msc_batch_size)"""
        nb = C.c_size_t(0); nd = C.c_uint32(0)
        self._chk(self.lib.dabphy_msc_batch_size(self.h, C.byref(nb), C.byref(nd)))
        return nb.value, nd.value

    def msc_drain_begin(self, buf=None, desc=None):
        """dabphy_msc_drain_begin: queue the bulk drain of every selected sub-channel of every ensemble into `buf` (uint8; page-locked from
        host_alloc for PCIe rate) and return (buf, desc records) at once; msc_drain_wait() completes it"""
        nb, nd = self.msc_batch_size()
        if buf is None:
            buf = np.zeros(max(nb, 1), np.uint8)
        if desc is None:
            desc = np.zeros(max(nd, 1), MSC_DESC_DTYPE)
        n = C.c_uint32(0)
        self._chk(self.lib.dabphy_msc_drain_begin(self.h, _p(desc), C.c_uint32(len(desc)), C.byref(n), _p(buf), C.c_size_t(buf.nbytes)))
        return buf, desc[:n.value]

    def msc_drain_wait(self):
        self._chk(self.lib.dabphy_msc_drain_wait(self.h))

    def msc_batch(self, buf=None, desc=None):
        """dabphy_get_msc_batch -> (buf, desc): service k's logical frames = buf[desc[k].offset:][: 4F * row_bytes].reshape(4F, row_bytes)[first_valid:n_rows]"""
        buf, desc = self.msc_drain_begin(buf, desc)
        self.msc_drain_wait()
        return buf, desc

    def superframes_ensemble(self, ensemble, idx, bitrate):
        """-> (events [4F], n_events, corrected superframes [n_slots][120*bitrate/8]) of one ensemble's sub-channel idx"""
        F = self._last
        ev = np.zeros(4 * F, SF_EVENT_DTYPE); ne = C.c_int32(0)
        sf = np.zeros((4 * F // 5 + 1, 15 * bitrate), np.uint8)
        self._chk(self.lib.dabphy_superframes_ensemble(self.h, ensemble, idx, _p(ev), C.byref(ne), _p(sf)))
        return ev, ne.value, sf

    def host_alloc(self, shape, dtype):
        """page-locked numpy array (dabphy_host_alloc); release with host_free(arr)"""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self._chk(self.lib.dabphy_host_alloc(C.c_size_t(nbytes), C.byref(p)))
        arr = np.frombuffer((C.c_uint8 * nbytes).from_address(p.value), dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", {}); self._pinned[arr.ctypes.data] = p
        return arr

    def host_free(self, arr):
        self.lib.dabphy_host_free.restype = None
        self.lib.dabphy_host_free(self._pinned.pop(arr.ctypes.data))

    def stream_open(self, ring_samples):
        self._chk(self.lib.dabphy_stream_open(self.h, C.c_uint64(ring_samples)))

    def stream_write(self, iq):
        iq = np.ascontiguousarray(iq, np.complex64).reshape(self.cfg.n_ensembles, -1)
        self._chk(self.lib.dabphy_stream_write(self.h, _p(iq), C.c_uint64(iq.shape[1])))

    def stream_write_raw(self, data, fmt):
        """data: [ensemble][n_samples][2] integers in the sample format fmt ("u8", "s8", "s16le", "s16be": CRAWFileFormat names)"""
        code = {"u8": 1, "s8": 2, "s16le": 3, "s16be": 4}[fmt]
        data = np.ascontiguousarray(data)
        bps = 2 if code <= 2 else 4
        n = data.nbytes // (self.cfg.n_ensembles * bps)
        self._chk(self.lib.dabphy_stream_write_raw(self.h, _p(data), C.c_uint64(n), code))

    def stream_write_raw_async(self, data, fmt):
        code = {"u8": 1, "s8": 2, "s16le": 3, "s16be": 4}[fmt]
        assert data.flags["C_CONTIGUOUS"]
        bps = 2 if code <= 2 else 4
        self._chk(self.lib.dabphy_stream_write_raw_async(self.h, _p(data), C.c_uint64(data.nbytes // (self.cfg.n_ensembles * bps)), code))

    def stream_read(self, ensemble, pos, n):
        out = np.zeros(n, np.complex64)
        self._chk(self.lib.dabphy_stream_read(self.h, ensemble, C.c_uint64(pos), C.c_uint64(n), _p(out)))
        return out

    def stream_commit(self):
        self._chk(self.lib.dabphy_stream_commit(self.h))

    def stream_consumed(self):
        self.lib.dabphy_stream_consumed.restype = C.c_uint64
        return int(self.lib.dabphy_stream_consumed(self.h))

    def process(self, n_frames):
        self._chk(self.lib.dabphy_process(self.h, n_frames))
        self._last = n_frames

    def frame_info(self):
        B, F = self.cfg.n_ensembles, self._last
        arr = (FrameInfo * (B * F))()
        self._chk(self.lib.dabphy_get_frame_info(self.h, arr))
        raw = np.frombuffer(arr, dtype=FRAME_INFO_RAW)
        out = np.zeros((B, F), dtype=[("pos", np.int64), ("frame_no", np.int64), ("start_index", np.int32), ("valid", np.int32),
                                      ("fine", np.int32), ("coarse", np.int32), ("snr", np.float32)])
        flat = out.reshape(-1)
        for dst, src in (("pos", "sample_pos"), ("frame_no", "frame_no"), ("start_index", "start_index"), ("valid", "valid"), ("fine", "fine_corrector"),
                         ("coarse", "coarse_corrector"), ("snr", "snr")):
            flat[dst] = raw[src]
        return out

    def fibs(self):
        B, F = self.cfg.n_ensembles, self._last
        fib = np.zeros((B, F, 12, 32), np.uint8); ok = np.zeros((B, F, 12), np.uint8)
        self._chk(self.lib.dabphy_get_fibs(self.h, _p(fib), _p(ok)))
        return fib, ok

    def fibs_host(self):
        """the same, as read-only views of the library's page-locked host copies (valid until the next process())"""
        B, F = self.cfg.n_ensembles, self._last
        pf = C.POINTER(C.c_uint8)(); po = C.POINTER(C.c_uint8)()
        self._chk(self.lib.dabphy_get_fibs_host(self.h, C.byref(pf), C.byref(po)))
        fib = np.ctypeslib.as_array(pf, shape=(B * F * 384,)).reshape(B, F, 12, 32)
        ok = np.ctypeslib.as_array(po, shape=(B * F * 12,)).reshape(B, F, 12)
        return fib, ok

    def fic_ratio(self):
        r = np.zeros(self.cfg.n_ensembles, np.int32)
        self._chk(self.lib.dabphy_get_fic_ratio(self.h, _p(r)))
        return r

    def msc(self, idx):
        """-> (out [B][4F][bytes], first_valid [B]); rows [first_valid[b], self.msc_rows[b]) are the batch's logical frames"""
        B, F = self.cfg.n_ensembles, self._last
        out = np.zeros((B, 4 * F, self._sub_bytes[idx]), np.uint8); fv = np.zeros(B, np.int32); self.msc_rows = np.zeros(B, np.int32)
        self._chk(self.lib.dabphy_get_msc(self.h, idx, _p(out), C.c_size_t(out.nbytes), _p(fv), _p(self.msc_rows)))
        return out, fv

    def config(self):
        c = Config(C.sizeof(Config)); self._chk(self.lib.dabphy_get_config_v2(self.h, C.byref(c))); return c

    def demod_chunk(self):
        return self.config().demod_chunk

    def set_options(self, fft_placement=2, freqsync_method=2, disable_coarse=False):
        """RadioReceiver::setReceiverOptions at run time; returns True when the change restarted the synchroniser"""
        r = C.c_int32(0)
        self._chk(self.lib.dabphy_set_options(self.h, fft_placement, freqsync_method, int(disable_coarse), C.byref(r)))
        return bool(r.value)

    def ratio_lag(self):
        """(frames whose coarse-corrector decision used a stale FIC ratio, frame number of the first one or -1) per ensemble"""
        n = np.zeros(self.cfg.n_ensembles, np.int32); f = np.zeros(self.cfg.n_ensembles, np.int64)
        self._chk(self.lib.dabphy_get_ratio_lag(self.h, _p(n), _p(f)))
        return n, f

    def ratio_lag_effect(self):
        """(stale coarse-corrector decisions that can have changed anything, frame number of the first one or -1) per ensemble"""
        n = np.zeros(self.cfg.n_ensembles, np.int32); f = np.zeros(self.cfg.n_ensembles, np.int64)
        self._chk(self.lib.dabphy_get_ratio_lag_effect(self.h, _p(n), _p(f)))
        return n, f

    def scan_stats(self):
        a = np.zeros(self.cfg.n_ensembles, np.int32); f = np.zeros(self.cfg.n_ensembles, np.int32)
        self._chk(self.lib.dabphy_get_scan_stats(self.h, _p(a), _p(f)))
        return a, f

    def fibs_device(self):
        """(fib [B][F][12][32], crc_ok [B][F][12]) as torch uint8 tensors aliasing the library's HBM buffers (valid until the next
        process()); for device-side consumers such as the RCCL gather of the multi-GPU bench"""
        import torch
        B, F = self.cfg.n_ensembles, self._last
        pf = C.c_void_p(); po = C.c_void_p()
        self._chk(self.lib.dabphy_get_fibs_device(self.h, C.byref(pf), C.byref(po)))

        class _Dev:
            def __init__(self, ptr, shape):
                self.__cuda_array_interface__ = {"shape": shape, "typestr": "|u1", "data": (ptr, False), "version": 2}
        dev = "cuda:%d" % self.cfg.device
        return (torch.as_tensor(_Dev(pf.value, (B, F, 12, 32)), device=dev), torch.as_tensor(_Dev(po.value, (B, F, 12)), device=dev))

    def impulse_response(self):
        B, F = self.cfg.n_ensembles, self._last
        out = np.zeros((B, F, 2048), np.float32)
        self._chk(self.lib.dabphy_get_impulse_response(self.h, _p(out)))
        return out

    def null_symbols(self):
        out = np.zeros((self.cfg.n_ensembles, self._last, 2656), np.complex64)
        self._chk(self.lib.dabphy_get_null_symbols(self.h, _p(out)))
        return out

    def constellation(self):
        B, F = self.cfg.n_ensembles, self._last
        out = np.zeros((B, F, 1200), np.complex64)
        self._chk(self.lib.dabphy_get_constellation(self.h, _p(out)))
        return out

    def soft_bits(self, ens, frame):
        out = np.zeros((75, 3072), np.int8)
        self._chk(self.lib.dabphy_get_soft_bits(self.h, ens, frame, _p(out)))
        return out


class FrameInfo(C.Structure):
    _fields_ = [("sample_pos", C.c_int64), ("frame_no", C.c_int64), ("start_index", C.c_int32), ("valid", C.c_int32),
                ("fine_corrector", C.c_int32), ("coarse_corrector", C.c_int32), ("snr", C.c_float)]


def _rs_superframes(self, sf, s_per_sf):
    sf = np.ascontiguousarray(sf, np.uint8).reshape(-1, 120 * s_per_sf).copy()
    n = sf.shape[0]; corr = np.zeros(n, np.int32); unc = np.zeros(n, np.int32)
    self._chk(self.lib.dabphy_rs_superframes(self.h, _p(sf), s_per_sf, n, _p(corr), _p(unc)))
    return sf, corr, unc


def _rs_decode_msc(self, subch_index, first_cif):
    B = self.cfg.n_ensembles
    fc = np.ascontiguousarray(first_cif, np.int32).reshape(B)
    corr = np.zeros(B, np.int32); unc = np.zeros(B, np.int32)
    self._chk(self.lib.dabphy_rs_decode_msc(self.h, int(subch_index), _p(fc), _p(corr), _p(unc)))
    return corr, unc


def _set_profiling(self, on=True):
    self._chk(self.lib.dabphy_set_profiling(self.h, int(on)))


def _stage_times(self):
    ms = np.zeros(7, np.float32)
    self._chk(self.lib.dabphy_get_stage_times(self.h, _p(ms)))
    return dict(zip(["sync", "demod", "snr", "fic", "msc_gather", "msc_viterbi", "rs"], [float(v) for v in ms]))


DabPhy.rs_superframes = _rs_superframes
DabPhy.rs_decode_msc = _rs_decode_msc
DabPhy.set_profiling = _set_profiling
DabPhy.stage_times = _stage_times
