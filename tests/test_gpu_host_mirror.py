"""GPU variant of tests/test_host_mirror.py: GpuRadioReceiver linked against the real libdabphy_hip.so, compared with
the reference facade (both prebuilt in oracle/_ref, which travels to the GPU box)."""
import os

import pytest

import refapi as R
from test_host_mirror import compare_runs
from welle_io_amd import synth

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (R.have_ref() and os.path.exists(R.GPU_HIP_SO)), reason="oracle/_ref not prebuilt")]


@pytest.mark.parametrize("snr,cfo,delay,nf", [(22, 0, 0, 20), (14, 137, 700, 14), (18, 2300, 0, 14)])
def test_same_callbacks_as_reference_facade(gpu, snr, cfo, delay, nf):
    x, tx = synth.make_stream(nf, snr_db=snr, cfo_hz=cfo, delay=delay, return_tx=True, seed=5)
    subs = [tx.subchs[2], tx.subchs[11], tx.subchs[17]]
    a = R.receiver_run(x, subchs=subs)
    b = R.gpu_receiver_run(x, subchs=subs, lib=R.GPU_HIP_SO)
    compare_runs(a, b, len(subs))
    if cfo == 0:
        assert b["n_services"] >= 18      # the reference's FIBProcessor, fed with our FIBs, announced every service
