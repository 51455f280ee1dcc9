"""The throughput workload of BASELINE.json config 4/5 ("batch of 256 synthetic Mode-I ensembles per GPU"), built ONCE here so
that bench.py and the parity test of the benchmarked configuration (tests/test_gpu_bench_config.py) decode the very same signal
with the very same handle configuration.

Signal: `n_distinct` looping recordings of REC = 20 frames (bench.py: 20 x ceil(frames per step / 20), so that a step reads no sample twice) of the canonical ensemble (18 x 64 kbit/s DAB+ EEP-3A, RS-valid
payload; 80 CIFs = a whole number of superframes and of interleaver periods, transmitter run for one period first so that the loop
point is seamless), each ensemble = recording b % n_distinct with its own carrier offset (a multiple of RATE/N, so the loop stays
phase-continuous) and its own AWGN (sigma 0.02 per axis, SURVEY 8d throughput setting).  torch is used for the arithmetic only
(device "cuda" in bench.py, "cpu" in the GPU-less tests)."""
import numpy as np

from . import synth

REC_FRAMES = 20            # recordings are multiples of 20 frames (whole superframes and interleaver periods)


def rec_frames_for(frames_per_step):
    """the looping recording is at least as long as a batch: no sample is read twice inside one step (a 20-frame ring under a 32-frame
    batch let the Infinity Cache serve part of the second reads: the FFT stage looked 6 % faster than it is)"""
    return REC_FRAMES * max(1, -(-int(frames_per_step) // REC_FRAMES))
RATE = 2048000.0


def uep_subchannel(lib, subch_id, start_cu, bitrate, level, dabplus=True):
    """a short-form (UEP) synth.SubchannelCfg: table row, size and (L_i, PI_i) segments from the library's own protection helpers (pure
    host functions of the C ABI: no device, no handle)"""
    import ctypes as C
    from .capi import Protection
    p = Protection(); assert lib.dabphy_protection_uep(C.byref(p), bitrate, level) == 0
    for idx in range(64):
        size = C.c_int(0); lvl = C.c_int(0); br = C.c_int(0)
        assert lib.dabphy_uep_table_entry(idx, C.byref(size), C.byref(lvl), C.byref(br)) == 0
        if br.value == bitrate and lvl.value == level:
            segs = [(p.L[i], p.PI[i]) for i in range(4) if p.PI[i] > 0 and p.L[i] > 0]
            return synth.SubchannelCfg(subch_id, start_cu, bitrate, level=level, dabplus=dabplus, uep=(idx, size.value, segs))
    raise ValueError("no UEP table row for %d kbit/s level %d" % (bitrate, level))


# A multiplex as they are on air (bench.py's `hetero` leg, tests/test_gpu_bench_config.py): 15 sub-channels in 6 protection classes --
# (bit rate, EEP profile B?, level) or ("uep", bit rate, level) -- 128 kbit/s 3-A, 6 x 64 kbit/s 3-A, 4 x 48 kbit/s 2-A, 2 x 32 kbit/s
# 3-B, 80 kbit/s UEP-3, 8 kbit/s 3-A: code words of 192 .. 3072 bits, 676 of the 864 capacity units.  All carry DAB+ superframes
# (what the reference's SuperframeFilter, dabplus_decoder.cpp:50-213, is fed): the filter runs on every one of them as in the headline.
HETERO_LAYOUT = [(128, False, 3)] + [(64, False, 3)] * 6 + [(48, False, 2)] * 4 + [(32, True, 3)] * 2 + [("uep", 80, 3), (8, False, 3)]


def hetero_subchannels(lib, layout=None):
    subchs = []; cu = 0
    for i, ent in enumerate(layout or HETERO_LAYOUT):
        sc = uep_subchannel(lib, i + 1, cu, ent[1], ent[2]) if ent[0] == "uep" else synth.SubchannelCfg(i + 1, cu, ent[0], ent[1], ent[2])
        subchs.append(sc); cu += sc.size_cu
    assert cu <= 864, cu
    return subchs


def subchannels_to_json(subchs):
    """plain-data form of a sub-channel list (for a CPU receiver in another process: tests/cpu_baseline_worker.py)"""
    return [dict(subch_id=s.subch_id, start_cu=s.start_cu, bitrate=s.bitrate, profile_b=bool(s.profile_b), level=s.level, dabplus=bool(s.dabplus),
                 uep=None if s.uep is None else [s.uep[0], s.uep[1], [list(x) for x in s.uep[2]]]) for s in subchs]


def subchannels_from_json(rows):
    return [synth.SubchannelCfg(r["subch_id"], r["start_cu"], r["bitrate"], r["profile_b"], r["level"], r["dabplus"],
                                uep=None if r["uep"] is None else (r["uep"][0], r["uep"][1], [tuple(x) for x in r["uep"][2]])) for r in rows]


def dev_protection(dev, s):
    """device protection record of a synth.SubchannelCfg"""
    return dev.protection_uep(s.bitrate, s.level) if s.uep is not None else dev.protection_eep(s.bitrate, s.profile_b, s.level)


def random_layout(lib, rng, n_min=3, n_max=14, dabplus=True):
    """a random multiplex as tools/sweep_multiplex.py draws them: n_min .. n_max sub-channels, EEP profile A at 8 .. 192 kbit/s,
    profile B at 32 .. 192 kbit/s (levels 1-4) and rows of the UEP table, packed from CU 0"""
    import ctypes as C
    uep_rows = []
    for idx in range(64):
        size = C.c_int(0); lvl = C.c_int(0); br = C.c_int(0)
        if lib.dabphy_uep_table_entry(idx, C.byref(size), C.byref(lvl), C.byref(br)) == 0 and 0 < br.value <= 192 and (not dabplus or br.value % 8 == 0):
            uep_rows.append((br.value, lvl.value))
    subchs = []; cu = 0
    want = int(rng.randint(n_min, n_max + 1))
    for sid in range(1, want + 1):
        for _ in range(8):                                 # a few draws until one fits the 864 capacity units
            kind = rng.rand()
            if kind < 0.2:
                br, lvl = uep_rows[int(rng.randint(len(uep_rows)))]
                sc = uep_subchannel(lib, sid, cu, br, lvl, dabplus=dabplus)
            elif kind < 0.45:
                sc = synth.SubchannelCfg(sid, cu, int(rng.choice([32, 64, 96, 128, 192])), True, int(rng.randint(1, 5)), dabplus=dabplus)
            else:
                sc = synth.SubchannelCfg(sid, cu, int(rng.choice([8, 16, 24, 32, 40, 48, 56, 64, 72, 80, 96, 112, 128, 160, 192])), False, int(rng.randint(1, 5)), dabplus=dabplus)
            if cu + sc.size_cu <= 864:
                subchs.append(sc); cu += sc.size_cu
                break
    return subchs


def mixed_layouts(lib, seed=2024):
    """The multiplexes of the `mixed_layouts` workload (bench.py extras, tests): a batch of INDEPENDENT ensembles -- every receiver of
    the reference selects its own services (msc-handler.cpp:61-127) --: the canonical one (18 x 64 kbit/s), the heterogeneous one
    (HETERO_LAYOUT), two random ones -- of the second the receiver selects every other service --, and the canonical signal again from
    which the receiver selects nothing (FIC only).
    -> (transmitted sub-channels per recording, selected sub-channels per recording)"""
    rng = np.random.RandomState(seed)
    tx = [synth.default_subchannels(), hetero_subchannels(lib), random_layout(lib, rng), random_layout(lib, rng), synth.default_subchannels()]
    sel = [list(tx[0]), list(tx[1]), list(tx[2]), list(tx[3][::2]), []]
    return tx, sel


def make_base_streams(n_distinct, n_frames=REC_FRAMES, seed0=0, subchs=None):
    """subchs: one sub-channel list for every recording, or a list of n_distinct lists (a different multiplex per recording)"""
    out, txs = [], []
    per_stream = subchs is not None and len(subchs) > 0 and isinstance(subchs[0], (list, tuple))
    for e in range(n_distinct):
        tx = synth.EnsembleTx(eid=0x1000 + seed0 + e, subchs=subchs[e] if per_stream else subchs, seed=seed0 + e, payload_fn=synth.dabplus_payload_fn(4 * n_frames, seed0 + e))
        for _ in range(n_frames):
            tx.next_frame()
        frames = [tx.next_frame() for _ in range(n_frames)]
        out.append(np.concatenate(frames).astype(np.complex64))
        txs.append(tx)
    return np.stack(out), txs


def make_batch(B, rank=0, n_distinct=4, cfo_max_hz=60.0, sigma=0.02, device="cuda", base=None, rec_frames=REC_FRAMES):
    """-> (iq [B][N] complex64 torch tensor on `device`, per-ensemble carrier offsets in Hz, base recordings, their transmitters)"""
    import torch
    if base is None:
        base = make_base_streams(n_distinct, rec_frames, seed0=100 * rank)
    base_np, txs = base
    N = base_np.shape[1]
    gbase = torch.from_numpy(base_np).to(device)
    gen = torch.Generator(device=device); gen.manual_seed(1234 + rank)
    iq = torch.empty((B, N), dtype=torch.complex64, device=device)
    rs = np.random.RandomState(4321 + rank)
    cfo_hz = np.round(rs.uniform(-cfo_max_hz, cfo_max_hz, B) * N / RATE) * RATE / N
    n_idx = torch.arange(N, device=device, dtype=torch.float64)
    for b in range(B):
        noise = torch.randn((N, 2), generator=gen, device=device, dtype=torch.float32) * sigma
        rot = torch.polar(torch.ones_like(n_idx), n_idx * (2.0 * np.pi * cfo_hz[b] / RATE)).to(torch.complex64)
        iq[b] = gbase[b % base_np.shape[0]] * rot + torch.view_as_complex(noise)
    return iq, cfo_hz, base_np, txs


def resample_periodic(gx, n_out, ppm, taps=16, chunk=1 << 22):
    """torch form of synth.resample_ppm for a PERIODIC recording gx (a seamless loop): y[n] = x(n (1 + ppm 1e-6)) for n < n_out, the loop
    continued as far as needed -- a non-looping stream of any length as a receiver whose sampling clock is `ppm` parts per million fast
    sees it (band-limited interpolation, Hann-windowed sinc).  The PRS of frame k arrives k * 196608 * ppm * 1e-6 samples early."""
    import torch
    N0 = gx.shape[0]
    y = torch.empty(n_out, dtype=torch.complex64, device=gx.device)
    half = taps // 2
    for n0 in range(0, n_out, chunk):
        n = torch.arange(n0, min(n_out, n0 + chunk), device=gx.device, dtype=torch.float64)
        t = n * (1.0 + ppm * 1e-6)
        k0 = torch.floor(t)
        frac = (t - k0).to(torch.float32)
        k0 = k0.to(torch.int64)
        acc = torch.zeros(n.shape[0], dtype=torch.complex64, device=gx.device)
        for j in range(-half + 1, half + 1):
            u = frac - j
            w = torch.sinc(u) * (0.5 + 0.5 * torch.cos(np.pi * u / half))
            acc += gx[(k0 + j) % N0] * w
        y[n0:n0 + n.shape[0]] = acc
    return y


def make_channel_batch(B, n_frames, rank=0, n_distinct=4, cfo_max_hz=60.0, sigma=0.02, ppm=None, snr_db=None, device="cuda", base=None, rec_frames=REC_FRAMES,
                       amplitude=0.25):
    """The headline's ensembles as a receiver MEETS them (bench.py's `drift` and `low_snr` legs): ensemble b = recording b % n_distinct with
    its own carrier offset and its own noise, and
      ppm     = (lo, hi): a sampling-clock offset of its own, magnitude log-uniform in [lo, hi] ppm, random sign -- the window index of
                every frame moves (ofdm-processor.cpp:337-350), the wide synchroniser pass's prediction of an unmoved window fails;
                the stream is NOT a loop then: n_frames frames of the continued recording, resampled (resample_periodic);
      snr_db  = (lo, hi): its own signal-to-noise ratio, uniform in dB (instead of sigma): Reed-Solomon corrects, FIBs fail, the
                coarse corrector's FIC-ratio feedback acts (ofdm-processor.cpp:397-409).
    -> (iq [B][n_frames * 196608] on `device`, dict of the per-ensemble parameters, base recordings, their transmitters)"""
    import torch
    if base is None:
        base = make_base_streams(n_distinct, rec_frames, seed0=100 * rank)
    base_np, txs = base
    N0 = base_np.shape[1]
    N = int(n_frames) * 196608
    gbase = torch.from_numpy(base_np).to(device)
    gen = torch.Generator(device=device); gen.manual_seed(9234 + rank)
    rs = np.random.RandomState(7321 + rank)
    if ppm is None:
        assert N % N0 == 0 or N <= N0, "a looping stream is a whole number of recordings"
        cfo_hz = np.round(rs.uniform(-cfo_max_hz, cfo_max_hz, B) * N0 / RATE) * RATE / N0
        ppm_b = np.zeros(B)
    else:
        cfo_hz = rs.uniform(-cfo_max_hz, cfo_max_hz, B)
        ppm_b = np.exp(rs.uniform(np.log(ppm[0]), np.log(ppm[1]), B)) * rs.choice([-1.0, 1.0], B)
    snr_b = rs.uniform(snr_db[0], snr_db[1], B) if snr_db is not None else None
    sig_b = np.sqrt(amplitude ** 2 / (10 ** (snr_b / 10)) / 2) if snr_db is not None else np.full(B, sigma)
    iq = torch.empty((B, N), dtype=torch.complex64, device=device)
    n_idx = torch.arange(N, device=device, dtype=torch.float64)
    for b in range(B):
        x = gbase[b % base_np.shape[0]]
        if ppm is not None:
            x = resample_periodic(x, N, float(ppm_b[b]))
        elif N > N0:
            x = x.repeat(N // N0)
        else:
            x = x[:N]
        noise = torch.randn((N, 2), generator=gen, device=device, dtype=torch.float32) * float(sig_b[b])
        rot = torch.polar(torch.ones_like(n_idx), n_idx * (2.0 * np.pi * cfo_hz[b] / RATE)).to(torch.complex64)
        iq[b] = x * rot + torch.view_as_complex(noise)
        del noise, rot, x
    return iq, dict(cfo_hz=cfo_hz, ppm=ppm_b, snr_db=snr_b, sigma=sig_b), base_np, txs


def open_receiver(capi, lib_path, iq, F, subchs, device=0, pipeline_sync=1, demod_chunk=0, profiling=True, loop=True, decode_shape=0, sync_early=0, deferred_filter=False):
    """the handle bench.py times: batch geometry B x F, looping HBM-resident ring, coarse corrector enabled, no constellation / CIR
    taps, all sub-channels decoded, superframe filter inside process() (deferred_filter: the pass of a batch rides beside the NEXT batch's
    FFT stage, dabphy_set_auto_superframes(2); superframes_stats() then returns the totals of the batch before the last process())"""
    B, N = iq.shape
    dev = capi.DabPhy(n_ensembles=B, max_frames=F, device=device, lib_path=lib_path, want_constellation=False, want_impulse_response=False,
                      disable_coarse=False, pipeline_sync=pipeline_sync, demod_chunk=demod_chunk, decode_shape=decode_shape, sync_early=sync_early)
    if iq.is_cuda:
        dev.stream_bind_device(iq.data_ptr(), N, N, N, loop=loop)
    else:
        dev.stream_upload(iq.numpy(), loop=loop)
    if len(subchs) > 0 and isinstance(subchs[0], (list, tuple)):
        # a batch of independent ensembles: subchs[b] = the selection of ensemble b (dabphy_set_subchannels_ensemble)
        assert len(subchs) == B
        for b in range(B):
            dev.set_subchannels_ensemble(b, [(s.subch_id, s.start_cu, s.size_cu, dev_protection(dev, s)) for s in subchs[b]])
    else:
        dev.set_subchannels([(s.subch_id, s.start_cu, s.size_cu, dev_protection(dev, s)) for s in subchs])
    if profiling:
        dev.set_profiling(True)
    dev.set_auto_superframes(2 if deferred_filter else True)
    return dev
