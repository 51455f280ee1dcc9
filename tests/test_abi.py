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
