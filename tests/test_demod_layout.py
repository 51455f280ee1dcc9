"""The index algebra of k_demod's in-place FFT exchanges (csrc/fft2048.h, "k_demod's variant"): bijections, the per-wave halves the
LDS-DMA may overwrite without a barrier, and bank-conflict freedom by the rules of MI355X_MICROARCH.md -- checked on the CPU by
tools/layout/demod_inplace_layout.py (its assertions are the test; the kernel's closed-form addresses are restated there)."""
import os
import runpy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_inplace_exchange_layout(capsys):
    runpy.run_path(os.path.join(ROOT, "tools", "layout", "demod_inplace_layout.py"), run_name="__main__")
    out = capsys.readouterr().out
    assert "worst 1 -way" in out and "2 -way" not in out and out.strip().endswith("reads in round C")


def test_layout_constants_match_the_kernel_header():
    src = open(os.path.join(ROOT, "welle.io_amd", "csrc", "fft2048.h")).read()
    assert "constexpr int FFT_RAW_PITCH = 260;" in src
    chk = open(os.path.join(ROOT, "tools", "layout", "demod_inplace_layout.py")).read()
    assert "PITCH = 260" in chk
