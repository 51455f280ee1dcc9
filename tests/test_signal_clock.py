"""Batch mode's signal-time clock (welle.io_amd/host/signal_clock.h).  The reference's FIBProcessor forgets an SId that
was signalled once unless it is signalled again within about two seconds -- of WALL clock (fib-processor.cpp:284-328): a phantom
service from two mis-decoded FIBs far apart is never listed by a receiver running in real time, but is listed by anything that
decodes faster than the signal runs (the reference itself included).  GpuBatchReceiver feeds every ensemble's FIBProcessor (the
reference's unmodified source) a clock that runs with the ensemble's signal, so its verdicts are those of a real-time receiver at
any decode speed."""
import numpy as np
import pytest

import refapi as R
from welle_io_amd import synth

PHANTOM_FRAMES = (5, 60)            # 0.5 s and 5.8 s into the signal


def phantom_stream(seed=0, eid=0x1000):
    def extra(frame):               # FIG 0/2: SId 0x4321, one DAB+ audio component in sub-channel 1
        return [bytes([0x06, 0x02, 0x43, 0x21, 0x01, 0x3F, (1 << 2) | 0x02])] if frame in PHANTOM_FRAMES else []
    return synth.make_stream(70, eid=eid, subchs=synth.default_subchannels(2), seed=seed, snr_db=25, extra_figs_fn=extra)


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
def test_reference_verdict_depends_on_its_pace(oracle_built):
    """the behaviour to preserve: paced in real time the reference lists the two real services; free-running it lists the phantom too"""
    x = phantom_stream()
    assert R.ref_service_list_run(x, realtime=True) == (2, 2)
    assert R.ref_service_list_run(x, realtime=False) == (3, 3)


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
def test_batch_receiver_ages_services_by_signal_time(emu):
    xs = np.stack([phantom_stream(seed=1, eid=0x1001), phantom_stream(seed=2, eid=0x1002)])
    listed, detected = R.gpu_batch_services(xs, 5, 14, signal_clock=True, lib=R.GPU_EMU_SO)
    assert list(listed) == [2, 2] and list(detected) == [2, 2]


@pytest.mark.gpu
def test_batch_receiver_ages_services_by_signal_time_on_device(gpu):
    """6.7 s of signal decoded in milliseconds: with the signal clock the real-time verdict, without it the phantom is listed"""
    xs = np.stack([phantom_stream(seed=1, eid=0x1001), phantom_stream(seed=2, eid=0x1002), phantom_stream(seed=3, eid=0x1003)])
    listed, detected = R.gpu_batch_services(xs, 10, 7, signal_clock=True, lib=R.GPU_HIP_SO)
    assert list(listed) == [2, 2, 2] and list(detected) == [2, 2, 2]
    listed, detected = R.gpu_batch_services(xs, 10, 7, signal_clock=False, lib=R.GPU_HIP_SO)
    assert list(listed) == [3, 3, 3]
