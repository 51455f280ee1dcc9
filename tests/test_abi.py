"""The C-ABI library exports every symbol include/dabphy.h and include/dabphy_test.h declare (no compute calls: no GPU here), and the
product refuses to start without a gfx950 device instead of falling back to anything."""
import os
import re
import subprocess

import pytest

from conftest import GPU_LIB, ROOT


def declared_functions():
    fns = set()
    for name in ("dabphy.h", "dabphy_test.h"):
        src = open(os.path.join(ROOT, "include", name)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        fns |= set(re.findall(r"\b(dabphy_[a-z0-9_]+)\s*\(", src))
    return sorted(fns)


def test_header_symbols_exported():
    # (re)build incrementally so the check always sees the current sources
    subprocess.run(["make", "-j4"], cwd=os.path.join(ROOT, "welle.io_amd", "csrc"), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    out = subprocess.run(["nm", "-D", "--defined-only", GPU_LIB], capture_output=True, text=True, check=True).stdout
    exported = set(line.split()[-1] for line in out.splitlines() if line.strip())
    fns = declared_functions()
    assert len(fns) >= 10
    prod = open(os.path.join(ROOT, "include", "dabphy.h")).read()
    assert "dabphy_time_" not in re.sub(r"/\*.*?\*/", "", prod, flags=re.S) and "dabphy_selftest_" not in re.sub(r"/\*.*?\*/", "", prod, flags=re.S)      # test drivers stay out of the receiver API
    missing = [f for f in fns if f not in exported]
    assert not missing, "declared in include/*.h but not exported: %s" % missing


def test_no_cpu_fallback():
    """on a machine without a GPU, creating a handle on the product library must fail loudly"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from welle_io_amd import capi
    with pytest.raises(capi.DabPhyError):
        capi.DabPhy(lib_path=GPU_LIB)


def test_object_built_against_round_3_header(emu, tmp_path):
    """include/dabphy.h "ABI versioning": a host object compiled against ROUND 3's header (tests/abi/dabphy_r3.h: the unsized 48-byte
    dabphy_config, no decode_shape) and linked against today's library creates a handle -- the fields added since take their defaults --
    and reads its configuration back in ITS layout (the exported symbols dabphy_create / dabphy_get_config stay frozen to it; today's
    header maps those names to the sized _v2 entry points).  Run on the library's CPU execution model build: same host code."""
    exe = str(tmp_path / "abi_r3_client")
    emudir = os.path.join(ROOT, "tests", "hipemu")
    subprocess.run(["gcc", "-O1", "-I" + os.path.join(ROOT, "tests", "abi"), os.path.join(ROOT, "tests", "native", "abi_r3_client.c"),
                    "-L" + emudir, "-ldabphy_emu", "-Wl,-rpath," + emudir, "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.split() == ["ok", "2", "3", "1", "2", "1", "1", "15"], (out.returncode, out.stdout, out.stderr)


def test_sized_configuration(emu):
    """dabphy_create_v2: a SHORTER structure (an older caller) is zero-extended, a LONGER one (fields this library does not know) refused;
    dabphy_get_config_v2 writes at most what the caller's structure holds"""
    import ctypes as C
    from conftest import EMU_LIB
    from welle_io_amd import capi
    lib = capi.load_library(EMU_LIB)
    assert lib.dabphy_abi_version() == capi.ABI_VERSION
    lib.dabphy_struct_size.restype = C.c_size_t
    assert lib.dabphy_struct_size(0) == C.sizeof(capi.Config) and lib.dabphy_struct_size(3) == C.sizeof(capi.Subchannel) and lib.dabphy_struct_size(4) == C.sizeof(capi.Protection)
    assert lib.dabphy_struct_size(1) == C.sizeof(capi.FrameInfo) and lib.dabphy_struct_size(2) == capi.SF_EVENT_DTYPE.itemsize and lib.dabphy_struct_size(5) == capi.TII_DTYPE.itemsize and lib.dabphy_struct_size(6) == capi.MSC_DESC_DTYPE.itemsize
    assert lib.dabphy_struct_size(99) == 0

    class Padded(C.Structure):
        _fields_ = [("cfg", capi.Config), ("poison", C.c_int32 * 4)]
    p = Padded(); C.memset(C.byref(p), 0x7f, C.sizeof(p))
    C.memset(C.byref(p.cfg), 0, C.sizeof(capi.Config))
    p.cfg.n_ensembles = 1; p.cfg.max_frames = 2; p.cfg.fft_placement = 2; p.cfg.freqsync_method = 2
    h = C.c_void_p()
    # round 3's fields + the size member, no decode_shape, no sync_early: the poison behind them must not be read
    p.cfg.struct_size = C.sizeof(capi.Config) - 8; p.cfg.decode_shape = 0x7f7f7f7f; p.cfg.sync_early = 0x7f7f7f7f
    assert lib.dabphy_create_v2(C.byref(p), C.byref(h)) == 0
    got = capi.Config(C.sizeof(capi.Config))
    assert lib.dabphy_get_config_v2(h, C.byref(got)) == 0 and got.decode_shape == 0 and got.sync_early == 0 and got.max_frames == 2 and got.struct_size == C.sizeof(capi.Config)
    lib.dabphy_destroy(h)
    # round 5's structure (decode_shape, no sync_early)
    p.cfg.struct_size = C.sizeof(capi.Config) - 4; p.cfg.decode_shape = 1; h = C.c_void_p()
    assert lib.dabphy_create_v2(C.byref(p), C.byref(h)) == 0
    got = capi.Config(C.sizeof(capi.Config))
    assert lib.dabphy_get_config_v2(h, C.byref(got)) == 0 and got.decode_shape == 1 and got.sync_early == 0
    short = capi.Config(12); short.max_frames = 99
    assert lib.dabphy_get_config_v2(h, C.byref(short)) == 0 and short.struct_size == 12 and short.max_frames == 2 and short.device == 0 and short.fft_placement == 0
    lib.dabphy_destroy(h)
    # only the geometry: the synchroniser options take the reference's defaults
    p.cfg.struct_size = 16; h = C.c_void_p()
    assert lib.dabphy_create_v2(C.byref(p), C.byref(h)) == 0
    got = capi.Config(C.sizeof(capi.Config))
    assert lib.dabphy_get_config_v2(h, C.byref(got)) == 0 and got.fft_placement == 2 and got.freqsync_method == 2 and got.decode_shape == 0
    lib.dabphy_destroy(h)
    for bad in (C.sizeof(capi.Config) + 4, 8, 0, 18):
        p.cfg.struct_size = bad; h = C.c_void_p()
        assert lib.dabphy_create_v2(C.byref(p), C.byref(h)) == -2 and not h.value, bad


def test_integration_md_shows_the_seam_files():
    """INTEGRATION.md level 2 shows welle.io_amd/host/seams/*.cpp VERBATIM (the files oracle/Makefile builds and tests/test_level2_seams.py
    tests), not a restatement that can drift"""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sync_integration.py"), "--check"])
    assert r.returncode == 0, "INTEGRATION.md is out of date: run python tools/sync_integration.py"
