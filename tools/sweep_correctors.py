"""One-off confidence sweep (GPU box): random SNR / CFO / delay streams, every frame's (fine, coarse) correctors and FIBs from the
device against the oracle, with the share of frames that needed the ordered float sums.  python tools/sweep_correctors.py [n] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: F401,E402
from conftest import GPU_LIB  # noqa: E402
import refapi as R  # noqa: E402
from welle_io_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
B = 8
bad = 0; frames = 0; exact = 0; relock = 0
for it in range(0, n, B):
    xs, orc = [], []
    for b in range(B):
        snr = float(rng.choice([8, 12, 16, 22, 30])); cfo = float(rng.uniform(-900, 2400)) if rng.rand() < 0.5 else float(rng.uniform(-60, 60))
        x = synth.make_stream(15, snr_db=snr, cfo_hz=cfo, delay=int(rng.randint(0, 2000)), seed=int(rng.randint(1 << 30)), noise_seed=int(rng.randint(1 << 30)))
        xs.append(x); orc.append(R.orc_receiver_run(x))
    L = max(len(x) for x in xs)
    xs = [np.concatenate([x, np.zeros(L - len(x), np.complex64)]) for x in xs]
    d = capi.DabPhy(lib_path=os.environ.get("DABPHY_LIB", GPU_LIB), n_ensembles=B, max_frames=1, want_constellation=False)
    d.stream_upload(np.stack(xs))
    got = [[] for _ in range(B)]; fibs = [[] for _ in range(B)]
    for step in range(15):                                # one frame per call: the coarse corrector sees the FIC ratio as the reference does
        d.process(1)
        info = d.frame_info(); fb, ok = d.fibs()
        for b in range(B):
            for f in range(1):
                if info[b, f]["valid"] == 1:
                    got[b].append((int(info[b, f]["fine"]), int(info[b, f]["coarse"]))); fibs[b].append(np.concatenate([ok[b, f][:, None], fb[b, f]], axis=1))
    lost, ex = d.sync_stats()
    d.close()
    for b in range(B):
        # the zero padding differs from the oracle's end of stream only after the last whole frame
        k = min(len(got[b]), len(orc[b]["corr"]) - 1)
        same = got[b][:k] == [tuple(c) for c in orc[b]["corr"][:k]] and np.array_equal(np.array(fibs[b][:k]).reshape(-1, 33), orc[b]["fib"][:12 * k])
        frames += k; exact += int(ex[b])
        if not same and lost[b] == 0:
            bad += 1
            print("MISMATCH stream", it + b, got[b][:k], [tuple(int(v) for v in c) for c in orc[b]["corr"][:k]])
        elif not same:
            relock += 1                                   # differs after a loss of lock
            print("MISMATCH after loss of lock, stream", it + b)
print("streams %d  frames %d  frames settled by ordered sums %d  streams differing after a loss of lock %d  other mismatches %d" % (n, frames, exact, relock, bad))
