"""Host-side check of the computed oscillator (welle.io_amd/csrc/osc_exact.h): accuracy of osc_exp over every table index,
and -- for chains as the kernels run them -- that every sample the rounding test does not flag equals the reference's
table entry (float)cos/sin(2 pi i / INPUT_RATE) (ofdm-processor.cpp:93-95) bit for bit."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_osc_exact_chain_matches_table(tmp_path):
    exe = str(tmp_path / "osc_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "tests", "hipemu"), "-I", os.path.join(ROOT, "welle.io_amd", "csrc"),
                           "-o", exe, os.path.join(ROOT, "tests", "native", "osc_check.cpp")])
    r = json.loads(subprocess.check_output([exe, "8000"]).decode())
    assert r["mismatch"] == 0
    assert r["exp_max_err"] < 1e-15
    assert r["chain_max_err"] * 6 < r["margin"]          # the margin the rounding test uses covers the chain error 6x over
    assert r["hard"] < 1e-4 * r["samples"]                # flagged samples (re-read from the table) stay rare
    # k_demod's unchecked conversion: exact for every table entry outside the unsafe list, with the error the header's budget allows
    assert r["unchecked_mismatch"] == 0 and r["unchecked_samples"] > 3e6
    assert 4 <= r["unsafe_entries"] <= 64
    assert r["exp_max_err"] < 1.4e-16 * 1.01             # the budget's per-value figure
    assert r["unchecked_max_err"] < 2.4e-15 and 10 * r["unchecked_max_err"] < r["unsafe_dist"]


def test_osc_hazard_mask_against_brute_force(tmp_path):
    """FrameDesc::osc_hazard (osc_hazard_entry: one congruence per unsafe table entry) marks exactly the symbols whose 2048 samples
    read an unsafe entry: compared with walking every sample's phase, over random and adversarial (f = 0, multiples of 500/1000,
    phases ON unsafe entries) frames; a missed symbol would be a wrong soft bit waiting to happen"""
    exe = str(tmp_path / "osc_hazard_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "tests", "hipemu"), "-I", os.path.join(ROOT, "welle.io_amd", "csrc"),
                           "-o", exe, os.path.join(ROOT, "tests", "native", "osc_hazard_check.cpp")])
    r = json.loads(subprocess.check_output([exe]).decode())
    assert r["missed"] == 0 and r["extra"] == 0 and r["flagged"] > 1000 and r["symbols"] > 100000
