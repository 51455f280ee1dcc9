"""pytest configuration: the `gpu` marker, package loading (the package directory is named `welle.io_amd`,
which is not an importable identifier, so it is loaded by path as `welle_io_amd`) and library fixtures."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "welle.io_amd")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def load_package():
    if "welle_io_amd" in sys.modules:
        return sys.modules["welle_io_amd"]
    spec = importlib.util.spec_from_file_location("welle_io_amd", os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["welle_io_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


load_package()

# The CPU execution model runs a wavefront as 64 fibres that meet at every cross-lane operation: the state-parallel Viterbi kernel
# (k_viterbi_sp: a few exchanges per trellis step) costs it seconds per frame.  The GPU-less suite therefore lets the library's automatic
# choice fall on the lane-per-code-word kernel (the experiments build of tests/hipemu reads DABPHY_SP_MAX_CW; the product library reads no
# environment at all) and runs the state-parallel kernel in the tests that ask for it (decode_shape = 2); the automatic choice itself is
# covered on the device (every small-batch test of the -m gpu suite) and by tests/test_emu_product_build.py.
os.environ.setdefault("DABPHY_SP_MAX_CW", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


# Cases of the execution-model suite whose IDENTICAL parametrisation also runs on the device (tests/test_gpu_stream.py, same function in
# tests/parity_cases.py): on the GPU-less machine they are skipped unless DABPHY_FULL_CPU_SUITE=1 -- each function keeps at least two
# parametrisations here, the execution model runs on one core and the whole CPU suite should stay a matter of minutes.
CPU_QUICK_SKIP = {
    "tests/test_emu_stream.py::test_low_snr_batches_with_coarse_corrector[2-40-4-9]",
    "tests/test_emu_stream.py::test_low_snr_batches_with_coarse_corrector[3--1000-4-5]",
    "tests/test_emu_stream.py::test_exact_batch_mode[4-17400-5-13-1-None]",
    "tests/test_emu_stream.py::test_exact_batch_mode[3--1000-4-5-3-None]",
    "tests/test_emu_stream.py::test_exact_batch_mode_with_different_ensembles[3]",
    "tests/test_emu_stream.py::test_wide_synchroniser_pass[2]",
    "tests/test_emu_stream.py::test_service_changes_while_the_synchroniser_runs_ahead[3]",
    "tests/test_emu_stream.py::test_live_ring_in_batches[3-300-17-3]",
    "tests/test_emu_stream.py::test_superframe_filter_where_the_damage_falls[2-damage_q0]",
}


def pytest_collection_modifyitems(config, items):
    if os.environ.get("DABPHY_FULL_CPU_SUITE"):
        return
    skip = pytest.mark.skip(reason="the device twin runs the same parametrisation (-m gpu); DABPHY_FULL_CPU_SUITE=1 runs it here too")
    for item in items:
        if item.nodeid in CPU_QUICK_SKIP:
            item.add_marker(skip)


EMU_LIB = os.path.join(ROOT, "tests", "hipemu", "libdabphy_emu.so")
GPU_LIB = os.environ.get("DABPHY_LIB") or os.path.join(PKG_DIR, "libdabphy_hip.so")
ORC_LIB = os.path.join(ROOT, "oracle", "libdabphy_oracle.so")


def _make(args, cwd):
    subprocess.run(["make"] + args, cwd=cwd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)


@pytest.fixture(scope="session", autouse=True)
def oracle_built():
    """the C restatement (checker) -- built from source when missing"""
    if not os.path.exists(ORC_LIB):
        _make(["oracle"], os.path.join(ROOT, "oracle"))
    return ORC_LIB


@pytest.fixture(scope="session")
def emu():
    """kernel sources compiled for the CPU execution model of tests/hipemu (logic check only, never timed)"""
    _make(["-j8", "emu"], os.path.join(PKG_DIR, "csrc"))
    from welle_io_amd import capi
    d = capi.DabPhy(lib_path=EMU_LIB)
    yield d
    d.close()


@pytest.fixture(scope="session")
def gpu():
    """the product library on a real device; fails loudly if it is missing"""
    from welle_io_amd import capi
    assert os.path.exists(GPU_LIB), "libdabphy_hip.so not built: run __graft_entry__.build()"
    # torch's bundled HIP runtime must come up BEFORE the library brings up the system one: the other order leaves torch without
    # devices ("No HIP GPUs are available"), and tests/test_gpu_bench_config.py builds its batch with torch on the device
    import torch
    assert torch.cuda.is_available(), "no GPU visible to torch"
    torch.cuda.init()
    d = capi.DabPhy(lib_path=GPU_LIB)
    assert "gfx950" in d.device_name
    yield d
    d.close()
