"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref, built from /root/reference by oracle/Makefile).
Run in the authoring container only:  python tests/golden/make_golden.py
The vectors let machines without /root/reference (the GPU box) pin the oracle and the HIP path to reference outputs."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import conftest  # noqa: F401,E402  (loads the package)
import refapi as R  # noqa: E402
import parity_cases as P  # noqa: E402
from welle_io_amd import synth  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    assert R.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    rng = np.random.RandomState(2024)
    out = {}
    # Viterbi::deconvolve
    for nbits in (768, 1536):
        s = rng.randint(-128, 128, (3, 4 * (nbits + 6))).astype(np.int8)
        out["vit%d_in" % nbits] = s
        out["vit%d_out" % nbits] = np.stack([np.packbits(R.ref_viterbi(s[i], nbits)) for i in range(3)])
    # FicHandler
    x = synth.make_stream(3, snr_db=8, seed=77)
    frames = P.cut_frames(x, 2)
    # one frame as RAW u8 IQ ((b-128)/128 like raw_file.cpp:324-366) so the input is exact and small
    u8 = synth.to_u8(frames[0])
    f0 = ((u8.astype(np.float32) - 128.0) / 128.0).view(np.complex64)[None, :]
    soft, con, _ = R.ref_ofdm_decode_frames(f0)
    out["frame_u8"] = u8
    out["frame_soft_sha"] = np.frombuffer(sha(soft).encode(), np.uint8)
    out["frame_soft_head"] = soft[0, :4]
    out["frame_con"] = con
    fic_soft = soft[0, :3].reshape(9216)
    bits, ok, ratio = R.ref_fic_decode(fic_soft)
    out["fic_bits"] = np.packbits(bits, axis=1); out["fic_ok"] = ok; out["fic_ratio"] = np.array([ratio])
    # EEP / UEP deconvolve
    s = rng.randint(-128, 128, R.orc_prot_eep(64, 0, 3).n_in).astype(np.int8)
    out["eep64_3a_in"] = s; out["eep64_3a_out"] = np.packbits(R.ref_eep(64, 0, 3, s))
    s = rng.randint(-128, 128, R.orc_prot_uep(80, 1).n_in).astype(np.int8)
    out["uep80_1_in"] = s; out["uep80_1_out"] = np.packbits(R.ref_uep(80, 1, s))
    # Reed-Solomon superframes with errors (incl. uncorrectable)
    sfs = np.stack([synth.make_superframe(64, rng) for _ in range(4)])
    for k, ne in enumerate((0, 4, 30, 300)):
        pos = rng.choice(sfs.shape[1], ne, replace=False); sfs[k, pos] ^= rng.randint(1, 256, ne).astype(np.uint8)
    res = [R.ref_rs_superframe(sfs[k]) for k in range(4)]
    out["rs_in"] = sfs; out["rs_out"] = np.stack([r[0] for r in res]); out["rs_corr"] = np.array([r[1] for r in res]); out["rs_unc"] = np.array([r[2] for r in res])
    # end-to-end through RadioReceiver: synthetic stream (regenerated from seeds by the test; its hash guards generator drift)
    x, tx = synth.make_stream(12, snr_db=14, cfo_hz=137, delay=321, seed=31, return_tx=True)
    subs = [tx.subchs[1], tx.subchs[8]]
    r = R.receiver_run(x, subchs=subs)
    out["e2e_input_sha"] = np.frombuffer(sha(x).encode(), np.uint8)
    out["e2e_fib"] = r["fib"]; out["e2e_corr"] = r["corr"]
    out["e2e_msc0_sha"] = np.frombuffer(sha(np.frombuffer(r["msc"][0], np.uint8)).encode(), np.uint8)
    out["e2e_msc0_len"] = np.array([len(r["msc"][0])])
    out["e2e_msc0_head"] = np.frombuffer(r["msc"][0][:384], np.uint8)
    out["e2e_con_sha"] = np.frombuffer(sha(r["con"]).encode(), np.uint8)
    # SuperframeFilter::Feed (the real class): a 64 kbit/s stream that starts mid-superframe, with correctable and
    # uncorrectable byte errors, a broken access unit and a dropped frame
    r2 = np.random.RandomState(99)
    sf = [synth.make_superframe(64, r2) for _ in range(8)]
    sf[2][7 * 8 + 1] ^= 0x55
    for k in range(4):
        sf[3][(20 + k) * 8] ^= 0xA0 + k
    sf[4][300] ^= 0x0F                                     # inside an access unit, within RS capacity
    for k in range(7):
        sf[5][(30 + 3 * k) * 8 + 1] ^= 0x11 * (k + 1)
    frames = np.concatenate(sf).reshape(-1, 192)[3:]
    frames = np.concatenate([frames[:23], frames[24:]])
    ev, sfs = R.ref_superframe_run(frames)
    out["sf_frames"] = frames
    out["sf_events"] = np.array([[e[0], e[1], e[2], e[3], e[4], e[5], e[7]] + list(e[6]) + [0] * (7 - len(e[6])) for e in ev], np.int32)
    out["sf_corrected_sha"] = np.frombuffer(sha(np.concatenate(sfs)).encode(), np.uint8)
    # TIIDecoder (the real class, every pair analysed): two transmitters, 16 (NULL, PRS) pairs regenerated from seeds by the test
    x = synth.make_stream(17, snr_db=18, seed=61, noise_seed=62, tii=P.TII_NETWORKS[0])
    nul, prs = P.tii_pairs(x, 16)
    out["tii_input_sha"] = np.frombuffer(sha(np.concatenate([nul.ravel(), prs.ravel()])).encode(), np.uint8)
    out["tii_events"] = np.array([[e[0], e[1], e[2], e[3], int(np.float32(e[4]).view(np.int32))] for e in R.ref_tii_run(nul, prs)], np.int32)
    out["tii_pair0"] = np.concatenate([nul[0], prs[0]])       # one pair verbatim (its 5-fold repetition gives a platform-independent case)
    out["tii_pair0_events"] = np.array([[e[0], e[1], e[2], e[3], int(np.float32(e[4]).view(np.int32))]
                                        for e in R.ref_tii_run(np.stack([nul[0]] * 10), np.stack([prs[0]] * 10))], np.int32)
    np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"), **out)
    print("wrote", os.path.join(HERE, "reference_vectors.npz"), os.path.getsize(os.path.join(HERE, "reference_vectors.npz")), "bytes")


if __name__ == "__main__":
    main()
