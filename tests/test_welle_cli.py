"""welle-cli ITSELF (BASELINE configs 1-3), built from the reference's unmodified sources by oracle/Makefile: once with the reference
backend (welle-cli-ref, config 1: `welle-cli -f <RAW u8 IQ file> -D`, CPU) and once against welle.io_amd/host/dropin/radio-receiver.h,
where the class called RadioReceiver is our facade over the C ABI (welle-cli-gpu-emu here: the kernels in the tests/hipemu execution
model; welle-cli-gpu-hip under -m gpu: configs 2-3 on the device).  Both decode the same u8 IQ file with `-D` (all programmes, FIC dump
+ one .msc dump per service); the dumps must agree byte for byte.

welle-cli paces a file in real time and attaches the programme decoders a wall-clock delay after start (welle-cli.cpp:623-636), so
the two .msc dumps start at different logical frames: they are aligned on whole frames and compared over their overlap.  The file is
long enough that the run ends before CRAWFile rewinds (its rewind flushes unread samples, raw_file.cpp:313-318: not reproducible)."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import ROOT
from welle_io_amd import synth, workload

REF_DIR = os.path.join(ROOT, "oracle", "_ref")
T_F = 196608


def _make_file(path, seconds):
    """seamless periodic recording (20 frames, 2 labelled DAB+ services) tiled to `seconds`, as RAW u8 IQ"""
    subchs = synth.default_subchannels(2)
    base, txs = workload.make_base_streams(1, subchs=subchs, seed0=7)
    u8 = synth.to_u8(base[0])
    reps = int(np.ceil(seconds * 2048000 * 2 / len(u8)))
    with open(path, "wb") as f:
        for _ in range(reps):
            f.write(u8.tobytes())
    return subchs


def _run_cli(binary, iq_path, cwd, decode_s, timeout_s):
    """start welle-cli -f file -D in cwd; `decode_s` seconds after it attached the programme decoders, type '.' to quit"""
    p = subprocess.Popen([binary, "-f", iq_path, "-D", "-T"], cwd=cwd, stdin=subprocess.PIPE, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    return p


def _finish(p, decode_s, timeout_s):
    t0 = time.time(); err = []
    attached = None
    os.set_blocking(p.stderr.fileno(), False)
    while time.time() - t0 < timeout_s:
        try:
            chunk = p.stderr.read()
        except (BlockingIOError, TypeError):
            chunk = None
        if chunk:
            err.append(chunk)
            if attached is None and "Enter '.' to quit" in "".join(err):
                attached = time.time()
        if p.poll() is not None:
            break
        if attached is not None and time.time() - attached >= decode_s:
            p.stdin.write(".\n"); p.stdin.flush()
            p.wait(timeout=60)
            break
        time.sleep(0.05)
    else:
        p.kill()
        raise AssertionError("welle-cli did not finish:\n" + "".join(err)[-3000:])
    assert p.returncode == 0 and attached is not None, "".join(err)[-3000:]
    return "".join(err)


def _compare_runs(tmp_path, gpu_binary, decode_s_ref, decode_s_gpu):
    iq = str(tmp_path / "rec.u8.iq")
    subchs = _make_file(iq, 40.0)
    d_ref = tmp_path / "ref"; d_gpu = tmp_path / "gpu"; d_ref.mkdir(); d_gpu.mkdir()
    p_ref = _run_cli(os.path.join(REF_DIR, "welle-cli-ref"), iq, str(d_ref), decode_s_ref, 120)
    p_gpu = _run_cli(gpu_binary, iq, str(d_gpu), decode_s_gpu, 120)
    e_gpu = _finish(p_gpu, decode_s_gpu, 200)
    e_ref = _finish(p_ref, 0.1, 200)
    assert "End of file, restarting" not in e_ref + e_gpu, "the run reached the rewind point: dumps are not comparable"
    # FIC: every FIB that passed its CRC, from the first frame on
    f_ref = (d_ref / "dump.fic").read_bytes(); f_gpu = (d_gpu / "dump.fic").read_bytes()
    n = min(len(f_ref), len(f_gpu)) // 32 * 32
    assert n >= 12 * 32 * 30, (len(f_ref), len(f_gpu))
    assert f_ref[:n] == f_gpu[:n], "dump.fic differs"
    # MSC: one dump per service, named after its label
    names = sorted(x for x in os.listdir(d_ref) if x.endswith(".msc"))
    assert names == sorted(x for x in os.listdir(d_gpu) if x.endswith(".msc")) and len(names) == len(subchs), (names, os.listdir(d_gpu))
    for nm in names:
        a = (d_ref / nm).read_bytes(); b = (d_gpu / nm).read_bytes()
        fb = subchs[0].frame_bytes
        na, nb = len(a) // fb, len(b) // fb
        assert na >= 20 and nb >= 20, (nm, na, nb)
        A = [a[i * fb:(i + 1) * fb] for i in range(na)]; B = [b[i * fb:(i + 1) * fb] for i in range(nb)]
        # align: the first frame of one dump somewhere in the other (the payload repeats every 80 frames: any match inside one period will do)
        best = 0
        for x, y in ((A, B), (B, A)):
            for off in range(min(len(y), 80)):
                k = 0
                while k < len(x) and off + k < len(y) and x[k] == y[off + k]:
                    k += 1
                if k == min(len(x), len(y) - off):
                    best = max(best, k)
        assert best >= 16, "%s: the dumps share no run of frames (ref %d frames, gpu %d)" % (nm, na, nb)
    return e_ref, e_gpu


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_DIR, "welle-cli-ref")), reason="oracle/_ref not built (needs /root/reference)")
def test_welle_cli_reference_vs_emu_backend(emu, tmp_path):
    _compare_runs(tmp_path, os.path.join(REF_DIR, "welle-cli-gpu-emu"), 4.0, 12.0)


@pytest.mark.gpu
def test_welle_cli_reference_vs_gpu_backend(gpu, tmp_path):
    assert os.path.exists(os.path.join(REF_DIR, "welle-cli-gpu-hip")), "oracle/_ref/welle-cli-gpu-hip must travel with the snapshot"
    e_ref, e_gpu = _compare_runs(tmp_path, os.path.join(REF_DIR, "welle-cli-gpu-hip"), 4.0, 4.0)
