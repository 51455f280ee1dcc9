"""Not a test: random SERVICE CHANGES on DAB+ ensembles, superframe filter included.  Every trial: 2 ... 4 ensembles of the canonical DAB+
multiplex (18 x 64 kbit/s, RS-valid superframes, each ensemble its own data, offset, delay), 1 ... 6 frames per call, any Viterbi decoder;
every ensemble starts with a random selection of its services and changes it at random between calls (a service joins, leaves, comes back:
MscHandler::addSubchannel / removeSubchannel, msc-handler.cpp:61-127).  A SESSION = one uninterrupted stay of a service in its ensemble's
list.  For every session: the logical frames it delivered are consecutive CIFs starting on the 17th CIF after it joined (dab-audio.cpp:146-149)
and equal the oracle's frames of those CIFs; the superframe events and corrected superframes the library reported for it
(dabphy_superframes_ensemble, batch by batch, through every re-indexing the other services' changes cause) equal those of ONE
SuperframeFilter fed with exactly these frames (dabplus_decoder.cpp:50-213).  5.5 ... 9 dB: at the low end byte errors reach Reed-Solomon.
python tools/sweep_services.py [n_trials] [seed] [first trial]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: F401,E402
from conftest import GPU_LIB  # noqa: E402
import parity_cases as P  # noqa: E402
import refapi as R  # noqa: E402
from welle_io_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
first_trial = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # earlier trials only draw their random numbers (to repeat one trial of a sweep)
lib_path = os.environ.get("DABPHY_LIB", GPU_LIB)
tot_sessions = tot_rows = tot_events = tot_changes = tot_corrected = 0
for it in range(n):
    B = int(rng.randint(2, 5)); F = int(rng.choice([1, 2, 3, 4, 6])); shape = int(rng.randint(0, 4)); pipe = int(rng.choice([0, 0, 1, 3]))
    n_steps = max(5, int(np.ceil(36 / F))); nf = F * n_steps + 3
    snr = float(rng.choice([5.5, 6.0, 7.0, 9.0]))
    par = [dict(cfo_hz=float(rng.uniform(-80, 80)), delay=int(rng.randint(0, 900)), seed=int(rng.randint(1 << 30)), pseed=int(rng.randint(1000))) for e in range(B)]
    pool = [sorted(rng.choice(18, int(rng.randint(2, 6)), replace=False).tolist()) for e in range(B)]      # the services an ensemble's listener ever asks for
    if it < first_trial:
        for e in range(B):
            rng.choice(pool[e], int(rng.randint(0, len(pool[e]) + 1)), replace=False)
        for step in range(1, n_steps):
            for e in range(B):
                if rng.rand() < 0.4:
                    rng.choice(pool[e], int(rng.randint(0, len(pool[e]) + 1)), replace=False)
        continue
    xs, txs = [], []
    for e in range(B):
        x, tx = synth.make_stream(nf, eid=0x6000 + 16 * it + e, snr_db=snr, cfo_hz=par[e]["cfo_hz"], delay=par[e]["delay"], return_tx=True,
                                  seed=par[e]["seed"], payload_fn=synth.dabplus_payload_fn(80, par[e]["pseed"]), noise_seed=78)
        xs.append(x); txs.append(tx)
    nmin = min(len(x) for x in xs); xs = [x[:nmin] for x in xs]
    orc = [R.orc_receiver_run(xs[e], subchs=[txs[e].subchs[i] for i in pool[e]]) for e in range(B)]
    want = [{i: np.frombuffer(bytes(orc[e]["msc"][k]), np.uint8).reshape(-1, txs[e].subchs[i].frame_bytes) for k, i in enumerate(pool[e])} for e in range(B)]

    def pick(e):
        k = int(rng.randint(0, len(pool[e]) + 1))
        return sorted(rng.choice(pool[e], k, replace=False).tolist())
    d = capi.DabPhy(lib_path=lib_path, n_ensembles=B, max_frames=F, want_constellation=False, want_impulse_response=False, decode_shape=shape, pipeline_sync=pipe)
    sub = lambda s: (s.subch_id, s.start_cu, s.size_cu, P.dev_prot(d, s))
    sel = [pick(e) for e in range(B)]
    open_ = {}                                      # (ensemble, service) -> session: rows, cifs, ev, sf, join (CIF count when it joined; None = learnt at the next batch)
    done = []

    def start(e, i, first):
        open_[(e, i)] = dict(e=e, i=i, rows=[], cifs=[], ev=[], sf=[], join=0 if first else None)
    for e in range(B):
        for i in sel[e]:
            start(e, i, True)
    try:
        d.stream_upload(np.stack(xs))
        for e in range(B):
            d.set_subchannels_ensemble(e, [sub(txs[e].subchs[i]) for i in sel[e]])
        for step in range(n_steps):
            if step:
                for e in range(B):
                    if rng.rand() < 0.4:
                        new = pick(e)
                        for i in set(sel[e]) - set(new):
                            done.append(open_.pop((e, i)))
                        for i in set(new) - set(sel[e]):
                            start(e, i, False)
                        sel[e] = new; tot_changes += 1
                        d.set_subchannels_ensemble(e, [sub(txs[e].subchs[i]) for i in sel[e]])
            d.process(F)
            info = d.frame_info()
            if not (info["valid"] == 1).any():
                break
            # (a slot whose window search fails yields no frame: the receiver's CIF count is the count of CIFs it received, here and in the oracle)
            for e in range(B):
                c0 = 4 * int(info[e, 0]["frame_no"])
                for idx, i in enumerate(sel[e]):
                    G = open_[(e, i)]; sc = txs[e].subchs[i]
                    if G["join"] is None:
                        G["join"] = c0
                    m, fv, nr = d.msc_ensemble(e, idx)
                    base_row = len(G["rows"])
                    for r in range(fv, nr):
                        G["rows"].append(m[r].copy()); G["cifs"].append(c0 + r)
                    ev, ne, sf = d.superframes_ensemble(e, idx, sc.bitrate)
                    for k in range(ne):
                        v = ev[k]
                        G["ev"].append((base_row + int(v["cif"]) - fv, int(v["corrected"]), int(v["uncorrectable"]), int(v["sync"]), int(v["format"]) if v["sync"] else 0, int(v["num_aus"]) if v["sync"] else 0,
                                        tuple(int(q) for q in v["au_start"][:v["num_aus"] + 1]) if v["sync"] else (), int(v["au_crc_ok"]) if v["sync"] else 0))
                        if v["sync"]:
                            G["sf"].append(sf[v["sf_slot"]].copy())
    finally:
        d.close()
    done += list(open_.values())
    for G in done:
        e, i = G["e"], G["i"]; tag = "trial %d ensemble %d service %d" % (it, e, txs[e].subchs[i].subch_id)
        if not G["rows"]:
            assert not G["ev"], tag + ": events without frames"
            continue
        first = G["cifs"][0]
        assert first == max(16, G["join"] + 16), tag + ": first frame on CIF %d, joined at %d" % (first, G["join"])
        assert G["cifs"] == list(range(first, first + len(G["cifs"]))), tag + ": a CIF is missing or repeated"
        fr = want[e][i][first - 16:first - 16 + len(G["rows"])]
        assert len(fr) == len(G["rows"]) and np.array_equal(np.stack(G["rows"]), fr), tag + ": bytes differ from the oracle's"
        eo, so = R.orc_superframe_run(fr)
        assert G["ev"] == eo[:len(G["ev"])] and len(G["ev"]) >= len(eo) - 1, tag + ": superframe events differ (%d reported, the filter fed with the same frames gives %d)" % (len(G["ev"]), len(eo))
        assert len(G["sf"]) <= len(so) and all(np.array_equal(G["sf"][k], so[k]) for k in range(len(G["sf"]))), tag + ": corrected superframes differ"
        tot_sessions += 1; tot_rows += len(G["rows"]); tot_events += len(G["ev"]); tot_corrected += sum(v[1] for v in G["ev"])
    print("trial %3d  %d ensembles  %d frames per call x %d steps  schedule %d  decode_shape %d  %.1f dB  sessions %d: equal" % (it, B, F, n_steps, pipe, shape, snr, len(done)), flush=True)
print("trials %d  sessions %d  logical frames %d  superframe events %d (bytes corrected by Reed-Solomon %d)  selection changes %d  mismatches 0" % (n, tot_sessions, tot_rows, tot_events, tot_corrected, tot_changes))
