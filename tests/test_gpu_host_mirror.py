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


def test_batch_receiver_feeds_one_fibprocessor_per_ensemble(gpu):
    """GpuBatchReceiver on the device: three ensembles in lock step, each parsed by its own reference FIBProcessor"""
    import numpy as np
    streams, eids = [], [0x10A1, 0x20B2, 0x30C3]
    for e, (eid, cfo) in enumerate(zip(eids, (0, 120, -80))):
        streams.append(synth.make_stream(9, eid=eid, snr_db=20, cfo_hz=cfo, seed=40 + e))
    eid, listed, ok, detected, n_tii = R.gpu_batch_run(np.stack(streams), 4, 2, lib=R.GPU_HIP_SO)
    assert list(eid) == eids and (listed == 18).all() and (ok >= 48).all() and (ok % 12 == 0).all()


def test_node_receiver_shards_ensembles_over_devices(gpu):
    """GpuNodeReceiver on the device: five ensembles over two shards (both on this box's one GPU: two handles, two host threads
    decoding concurrently) = one GpuBatchReceiver over all five"""
    import numpy as np
    streams = [synth.make_stream(9, eid=0x1000 + 0x111 * e, snr_db=20, cfo_hz=[0, 120, -80, 33, -7][e], seed=60 + e) for e in range(5)]
    x = np.stack(streams)
    one = R.gpu_batch_run(x, 4, 2, lib=R.GPU_HIP_SO)
    eid, listed, ok, detected, shards = R.gpu_node_run(x, [0, 0], 4, 2, lib=R.GPU_HIP_SO)
    assert shards == 2 and list(eid) == [0x1000 + 0x111 * e for e in range(5)]
    assert list(listed) == list(one[1]) and list(ok) == list(one[2]) and list(detected) == list(one[3])
    assert (listed == 18).all() and (ok >= 48).all()


def test_facade_accepts_all_sync_options(gpu):
    import numpy as np
    x, tx = synth.make_stream(10, snr_db=18, cfo_hz=2300, delay=300, return_tx=True, seed=6)
    subs = [tx.subchs[0]]
    a = R.receiver_run(x, subchs=subs, fft_placement=1, freqsync=1)
    b = R.gpu_receiver_run(x, subchs=subs, lib=R.GPU_HIP_SO, fft_placement=1, freqsync=1)
    n = min(len(a["fib"]), len(b["fib"]))
    assert n >= len(a["fib"]) - 12 and n > 24 and np.array_equal(a["fib"][:n], b["fib"][:n])


def test_facade_reports_tii_measurements(gpu):
    """decodeTII through the façade on the device: onTIIMeasurement = the TIIDecoder restatement over the same frames"""
    import parity_cases as P
    x = synth.make_stream(13, snr_db=20, cfo_hz=-40, delay=210, seed=44, tii=P.TII_NETWORKS[0])
    o = R.orc_receiver_run(x, tii=True)
    b = R.gpu_receiver_run(x, lib=R.GPU_HIP_SO, tii=True)
    nfr = len(b["nul"])
    assert b["tii"] == [e for e in o["tii"] if e[0] < nfr] and len(b["tii"]) >= 2


def test_facade_through_a_dropout(gpu):
    """the drop-in receiver on the device through a loss of lock: same FIBs, impulse responses and null symbols as the reference facade"""
    import numpy as np
    T_F = 196608
    x = synth.make_stream(19, snr_db=18, cfo_hz=60, delay=200, seed=8).copy()
    x[9 * T_F + 50000:10 * T_F + 120000] = 0
    a = R.receiver_run(x)
    b = R.gpu_receiver_run(x, lib=R.GPU_HIP_SO)
    n = min(len(a["fib"]), len(b["fib"]))
    assert a["n_sync_false"] > 3 and n >= len(a["fib"]) - 12 and n >= 12 * 14 and np.array_equal(a["fib"][:n], b["fib"][:n])
    kk = min(len(a["cir"]), len(b["cir"]))
    assert kk >= len(a["cir"]) - 1 and np.array_equal(a["cir"][:kk].view(np.uint32), b["cir"][:kk].view(np.uint32))
    kn = min(len(a["nul"]), len(b["nul"]))
    assert kn >= len(a["nul"]) - 1 and np.array_equal(a["nul"][:kn].view(np.uint32), b["nul"][:kn].view(np.uint32))


def test_scan_mode_signal_presence(gpu):
    from test_host_mirror import scan_streams
    good, noise = scan_streams()
    assert R.gpu_scan_run(good, lib=R.GPU_HIP_SO) == [1]
    assert R.gpu_scan_run(noise, lib=R.GPU_HIP_SO) == [0]


def test_worker_failure_becomes_input_failure(gpu):
    """an exception on the facade's worker thread surfaces as onInputFailure() (ofdm-processor.cpp:492-499), on the device build too"""
    x = synth.make_stream(6, snr_db=20, seed=2)
    assert R.gpu_failing_input_run(x, 3 * 196608, lib=R.GPU_HIP_SO) == 1


def test_batch_receiver_decodes_each_ensembles_own_services(gpu):
    """GpuBatchReceiver on the device: three different multiplexes in one batch, every ensemble selecting (and re-selecting in
    mid-stream) its own services, each into its own reference DecoderAdapter: dumps and Reed-Solomon statistics = three reference
    RadioReceivers'"""
    from conftest import GPU_LIB
    from test_host_mirror import check_batch_receiver_services
    check_batch_receiver_services(GPU_LIB, R.GPU_HIP_SO)
