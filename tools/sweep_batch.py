"""One-off confidence sweep (GPU box) of the batch-mode machinery of round 2: random SNR / carrier offset / delay / batch depth /
schedule, every stream through tests/parity_cases.check_stream_vs_oracle -- FIBs, CRC flags, correctors, all soft bits, null
symbols, SNR reports and the MSC bytes of three sub-channels against the oracle, frame by frame; batch mode's documented deviation
(a coarse-corrector decision taken with a stale FIC ratio) is tolerated only from the frame the library itself reports -- and reports
how many frames came from the wide synchroniser pass and how many OFDM symbols took the unchecked / checked oscillator conversion.
python tools/sweep_batch.py [n_streams] [seed] [exact | channels | wild]
With `exact` the library's default is swept instead -- exact batch mode, replay armed -- at 2-8 dB, and NO tolerance is given: every frame
must equal the oracle's; the replayed batches are counted.
With `channels` (exact batch mode too, 6-20 dB) every stream also passes a random channel: one to three echoes with complex gains and delays
from -250 to 700 samples (pre-echoes, echoes beyond the guard interval), a sampling-clock offset of up to +-120 ppm, flat fading of
10-40 % at 2-12 Hz -- each with probability 1/2 --, and a random FFT placement method: the signals that break the wide pass' prediction.
With `wild` (exact batch mode, all four schedules) 10-13 dB at 150-300 Hz offset in batches of 8-16 frames: the fine corrector needs ten frames
to get there, the reference's coarse corrector meanwhile takes false alarms and throws the receiver out of lock -- batches that are decoded
twice AND lose lock inside (where round 5's sweep of independent ensembles found the history ring not put back: DESIGN.md section 7)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: F401,E402
from conftest import GPU_LIB  # noqa: E402
import parity_cases as P  # noqa: E402
from welle_io_amd import capi, synth  # noqa: E402
import refapi as R  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lib = os.environ.get("DABPHY_LIB", GPU_LIB)
chan = len(sys.argv) > 3 and sys.argv[3] == "channels"
wild = len(sys.argv) > 3 and sys.argv[3] == "wild"
exact = (len(sys.argv) > 3 and sys.argv[3] == "exact") or chan or wild


SHAPE = int(os.environ.get("SWEEP_DECODE_SHAPE", "0"))      # dabphy_config.decode_shape: 0 = the default (these small batches: state-parallel kernel), 1 = lane per code word (the fused kernel's 144- / 324-row builds at 2 ... 8 frames per call)


def factory(**kw):
    return capi.DabPhy(lib_path=lib, decode_shape=SHAPE, **kw)


frames = wide = fast = checked = lagged = effective = replayed = skipped = 0
for it in range(n):
    snr = float(rng.choice([2, 3, 4, 5, 6, 8])) if exact else float(rng.choice([10, 13, 16, 20, 25, 30]))
    cfo = float(rng.uniform(-60, 60)) if rng.rand() < 0.5 else float(rng.uniform(-450, 450))
    delay = int(rng.randint(0, 2000)); F = int(rng.choice([2, 3, 5, 8])); pipe = int(rng.choice([0, 1, 2])); seed = int(rng.randint(1 << 30))
    nf = int(rng.choice([18, 26, 34]))
    if exact and rng.rand() < 0.5:
        cfo = float(rng.choice([-2400, -1000, 300, 1500, 2300, 17400]))
    channel = None; placement = 2; desc = ""
    if wild:
        # drawn again until the REFERENCE loses lock in mid-stream (the oracle alone is cheap): only those streams reach the case
        while True:
            snr = float(rng.choice([10, 13])); cfo = float(rng.uniform(150, 300)) * (1 if rng.rand() < 0.5 else -1)
            F = int(rng.choice([8, 12, 16])); pipe = int(rng.choice([0, 1, 2, 3])); nf = int(rng.choice([34, 42, 50])); seed = int(rng.randint(1 << 30))
            if R.orc_receiver_run(synth.make_stream(nf, snr_db=snr, cfo_hz=cfo, delay=delay, seed=seed))["n_sync_false"] >= 2:
                break
            skipped += 1
    if chan:
        snr = float(rng.choice([6, 8, 10, 14, 20]))
        channel = {}
        if rng.rand() < 0.5:
            channel["echoes"] = [(int(rng.randint(-250, 700)), complex(rng.uniform(0.2, 0.9) * np.exp(1j * rng.uniform(0, 2 * np.pi)))) for _ in range(int(rng.randint(1, 4)))]
        if rng.rand() < 0.5:
            channel["ppm"] = float(rng.uniform(-120, 120))
        if rng.rand() < 0.5:
            channel["fade"] = (float(rng.uniform(0.1, 0.4)), float(rng.uniform(2, 12)))
        placement = int(rng.choice([0, 1, 2]))
        desc = "  placement %d  channel %s" % (placement, {k: (["%d:%.2f%+.2fj" % (d_, g.real, g.imag) for d_, g in v] if k == "echoes" else v) for k, v in channel.items()})
    logs, o, _ = P.check_stream_vs_oracle(factory, snr, cfo, delay, nf, False, B=2, F=F, pipeline_sync=pipe, seed=seed, ratio_lag_ok=not exact,
                                          fft_placement=placement, channel=channel or None)
    replayed += logs[0]["replayed"]
    L = logs[0]
    k = len(L["info"]); frames += k; wide += L["wide"][0]; fast += L["osc"][0]; checked += L["osc"][1]; lagged += int(L["ratio_lag"][0] > 0); effective += int(L["ratio_lag_effect"][0] > 0)
    print("stream %3d  snr %4.0f dB  cfo %7.1f Hz  delay %4d  F %d  schedule %d  frames %2d  from the wide pass %2d  oscillator symbols unchecked/checked %d/%d  ratio lag %s with an effect %s"
          % (it, snr, cfo, delay, F, pipe, k, L["wide"][0], L["osc"][0], L["osc"][1], L["ratio_lag"], L["ratio_lag_effect"]) + desc, flush=True)
if wild:
    print("(%d streams drawn in which the reference holds its lock: not run)" % skipped)
print("decode_shape %d  streams %d  frames %d (x 2 ensembles)  accepted from the wide pass %d  oscillator symbols unchecked %d / checked %d  streams with a reported stale-ratio decision %d (with an effect: %d)  batches decoded twice %d  mismatches 0"
      % (SHAPE, n, frames, wide, fast, checked, lagged, effective, replayed))
