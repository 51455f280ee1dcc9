"""The kernel sources compiled for the CPU execution model a SECOND time, configured exactly like the shipped library (`make emu-product`:
DEMOD_WAVES=3, no -DDABPHY_EXPERIMENTS, hence no getenv override anywhere): the stream / mixed-class / bench-configuration checks once more,
so that the GPU-less suite also covers the product's own build switches (every other CPU test runs the experiments build `make emu`)."""
import os
import subprocess

import pytest

import parity_cases as P
from conftest import PKG_DIR, ROOT
from welle_io_amd import capi

LIB = os.path.join(ROOT, "tests", "hipemu", "libdabphy_emu_product.so")


@pytest.fixture(scope="module")
def emu_product():
    subprocess.run(["make", "emu-product"], cwd=os.path.join(PKG_DIR, "csrc"), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    return LIB


def factory(**kw):
    return capi.DabPhy(lib_path=LIB, **kw)


def test_environment_is_not_read(emu_product, monkeypatch):
    """the experiments build obeys DABPHY_FUSED_MSC=0 (two-kernel decode: a gather stage appears); the product build must not"""
    monkeypatch.setenv("DABPHY_FUSED_MSC", "0")
    monkeypatch.setenv("DABPHY_SP_MAX_CW", "0")
    P.check_mixed_ensemble(lambda **kw: capi.DabPhy(lib_path=LIB, decode_shape=1, **kw), F=4, nf=11, expect_fused=True)


@pytest.mark.parametrize("snr,cfo,delay,nf,lockstep", [(13, 137, 1000, 8, True), (20, -400, 333, 8, False)])
def test_stream(emu_product, snr, cfo, delay, nf, lockstep):
    P.check_stream_vs_oracle(factory, snr, cfo, delay, nf, lockstep)


def test_benchmark_handle_configuration_small(emu_product):
    from welle_io_amd import workload
    P.check_bench_config(capi, LIB, 3, 3, 1, check_ens=[0, 2], n_steps=4, demod_chunk=25, device="cpu", subs_idx=(0, 17),
                         base=workload.make_base_streams(2, workload.REC_FRAMES, seed0=0), expect_chunk=25, decode_shape=1)
