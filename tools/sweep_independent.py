"""Not a test: random batches of INDEPENDENT ensembles through the streaming receiver on the GPU box -- what round 5's pair tables are for.
Every trial draws 2 ... 6 ensembles, each with its OWN random multiplex (workload.random_layout: 3 ... 12 sub-channels; EEP A / B, UEP,
8 ... 192 kbit/s), its own stream (SNR, carrier offset, delay), its own initial selection of services and its own schedule of changes
(before a step, with probability 1/3, an ensemble replaces its selection by another random subset: services join, leave and stay), a batch
depth (1 ... 12 frames per call) and a Viterbi kernel (dabphy_config.decode_shape 0 ... 3), and compares with the oracle, run once per
ensemble over all its sub-channels:
  * FIBs and CRC flags of every frame of every ensemble;
  * every logical frame every selected service delivers, by its CIF number: frame of CIF c = the oracle's frame c - 16 of that sub-channel;
  * a service that stays selected through a change delivers consecutive CIFs (nothing lost, nothing repeated); one that joins before the
    batch that starts at CIF c0 delivers from CIF c0 + 16 on (dab-audio.cpp:146-149).
No tolerance.  python tools/sweep_independent.py [n_trials] [seed] [first trial]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: F401,E402
from conftest import GPU_LIB  # noqa: E402
import parity_cases as P  # noqa: E402
import refapi as R  # noqa: E402
from welle_io_amd import capi, synth, workload  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0       # [n_trials] [seed] [first trial]: earlier trials only draw their random numbers (to repeat one trial of a sweep)
lib_path = os.environ.get("DABPHY_LIB", GPU_LIB)
lib = capi.load_library(lib_path)
tot_frames = tot_rows = tot_changes = 0
for it in range(n):
    B = int(rng.randint(2, 7)); F = int(rng.choice([1, 2, 3, 4, 5, 8, 12])); shape = int(rng.randint(0, 4))
    n_steps = max(4, int(np.ceil(30 / F))); nf = F * n_steps + 2
    layouts = [workload.random_layout(lib, rng, 3, 12, dabplus=False) for _ in range(B)]
    par = [dict(snr_db=float(rng.choice([10, 13, 18, 25])), cfo_hz=float(rng.uniform(-300, 300)), delay=int(rng.randint(0, 900)), seed=int(rng.randint(1 << 30))) for e in range(B)]
    if it < first:
        for e in range(B):
            k = int(rng.randint(0, len(layouts[e]) + 1)); rng.choice(len(layouts[e]), k, replace=False)
        for step in range(1, n_steps):
            for e in range(B):
                if rng.rand() < 1 / 3:
                    k = int(rng.randint(0, len(layouts[e]) + 1)); rng.choice(len(layouts[e]), k, replace=False)
        continue
    xs = [synth.make_stream(nf, eid=0x5000 + 16 * it + e, subchs=layouts[e], **par[e]) for e in range(B)]
    nmin = min(len(x) for x in xs); xs = [x[:nmin] for x in xs]
    orc = [R.orc_receiver_run(xs[e], subchs=layouts[e]) for e in range(B)]
    d = capi.DabPhy(lib_path=lib_path, n_ensembles=B, max_frames=F, want_constellation=False, want_impulse_response=False, decode_shape=shape)
    sub = lambda s: (s.subch_id, s.start_cu, s.size_cu, P.dev_prot(d, s))

    def pick(e):
        k = int(rng.randint(0, len(layouts[e]) + 1))
        return [layouts[e][i] for i in sorted(rng.choice(len(layouts[e]), k, replace=False))]
    sel = [pick(e) for e in range(B)]
    last_cif = {}                                   # (ensemble, subch_id) -> last CIF delivered while selected without interruption
    joined = {(e, s.subch_id): 0 for e in range(B) for s in sel[e]}      # -> CIF count of the ensemble when it joined (None: must be learnt)
    fibs = [[] for _ in range(B)]
    try:
        d.stream_upload(np.stack(xs))
        for e in range(B):
            d.set_subchannels_ensemble(e, [sub(s) for s in sel[e]])
        for step in range(n_steps):
            changed = set()
            if step:
                for e in range(B):
                    if rng.rand() < 1 / 3:
                        new = pick(e); old_ids = {s.subch_id for s in sel[e]}; new_ids = {s.subch_id for s in new}
                        for sid in old_ids - new_ids:
                            last_cif.pop((e, sid), None); joined.pop((e, sid), None)
                        for sid in new_ids - old_ids:
                            joined[(e, sid)] = None
                        sel[e] = new; changed.add(e); tot_changes += 1
                        d.set_subchannels_ensemble(e, [sub(s) for s in sel[e]])
            d.process(F)
            info = d.frame_info(); fb, ok = d.fibs()
            if not (info["valid"] == 1).any():
                break
            for e in range(B):
                for f in range(F):
                    if info[e, f]["valid"] == 1:
                        fibs[e].append((ok[e, f].copy(), fb[e, f].copy()))
                c0 = 4 * int(info[e, 0]["frame_no"])
                for idx, s in enumerate(sel[e]):
                    key = (e, s.subch_id)
                    if joined[key] is None:
                        joined[key] = c0
                    m, fv, nr = d.msc_ensemble(e, idx)
                    want = np.frombuffer(bytes(orc[e]["msc"][layouts[e].index(s)]), np.uint8).reshape(-1, s.frame_bytes)
                    for r in range(fv, nr):
                        c = c0 + r
                        assert c >= joined[key] + 16, "trial %d ensemble %d service %d: CIF %d delivered, selected at %d" % (it, e, s.subch_id, c, joined[key])
                        if key in last_cif:
                            assert c == last_cif[key] + 1, "trial %d ensemble %d service %d: CIF %d after %d" % (it, e, s.subch_id, c, last_cif[key])
                        else:
                            assert c == max(16, joined[key] + 16), "trial %d ensemble %d service %d: first CIF %d, selected at %d" % (it, e, s.subch_id, c, joined[key])
                        last_cif[key] = c
                        assert c - 16 < len(want) and np.array_equal(m[r], want[c - 16]), "trial %d ensemble %d service %d (%d kbit/s): frame of CIF %d differs" % (it, e, s.subch_id, s.bitrate, c)
                        tot_rows += 1
    finally:
        d.close()
    for e in range(B):
        # the device has been given n_steps * F frame slots of a stream of two frames more; the oracle all of the stream.  A slot yields no
        # frame while the receiver acquires or when the window search fails (dabphy_frame_info.valid 0 / 3: the next slot searches again),
        # so a noisy stream delivers fewer frames than slots; which frames it delivers is checked by the FIB comparison below, in order
        n_dev, n_orc = len(fibs[e]), len(orc[e]["fib"]) // 12
        k = min(n_dev, n_orc)
        assert n_dev <= n_orc and 2 * k >= n_steps * F, (it, e, n_dev, n_orc)
        if k < n_steps * F - F - 1:
            print("trial %3d ensemble %d: %d frames in %d slots (the oracle: %d of the %d sent)" % (it, e, n_dev, n_steps * F, n_orc, nf), flush=True)
        of = orc[e]["fib"][:12 * k].reshape(k, 12, 33)
        assert all(np.array_equal(fibs[e][i][0], of[i, :, 0]) and np.array_equal(fibs[e][i][1], of[i, :, 1:]) for i in range(k)), "trial %d ensemble %d: FIBs differ" % (it, e)
        tot_frames += k
    print("trial %3d  %d ensembles (%s sub-channels)  %2d frames per call x %d steps  decode_shape %d: equal" % (it, B, "/".join(str(len(l)) for l in layouts), F, n_steps, shape), flush=True)
print("trials %d  frames %d  logical frames compared %d  selection changes %d  mismatches 0" % (n, tot_frames, tot_rows, tot_changes))
