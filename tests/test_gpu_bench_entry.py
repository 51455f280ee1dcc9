"""bench.py's own entry points on the device, small sizes: the line's contract fields, the parity leg (GPU vs CPU receivers on the same
rows), the RCCL gather of device buffers with one rank, and `--gpus 2` starting its two ranks itself (on a 1-GPU box the two ranks
share the device and the collective falls back to gloo -- RCCL refuses two ranks on one GPU; with >= 2 GPUs it is the real thing)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(args, env_extra=None, timeout=900):
    env = dict(os.environ, DABPHY_BENCH_QUICK="1", **(env_extra or {}))      # (half the CPU baseline's sample, shorter facade streams: the contract is checked, not the figures)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_and_parity_leg(gpu):
    j = run_bench(["--gpus", "1", "--steps", "2", "--warmup", "2", "--ensembles", "8", "--frames", "10", "--no-alt-schedule"])
    assert j["n_gpus"] == 1 and j["rccl_ranks"] == 0 and j["value"] > 0 and j["higher_is_better"] is True
    for k in ("roofline", "roofline_viterbi", "cpu_baseline", "parity_check", "stages_ms"):
        assert k in j, k
    assert j["roofline"]["bound"] == "hbm" and 0 < j["roofline"]["frac"] < 1
    assert j["parity_check"]["fib_equal"] and j["parity_check"]["msc_equal"] and j["parity_check"]["frames"] >= 15
    assert j["parity_check"]["ranks_ok"] == 1 and len(j["parity_check"]["ensembles"]) >= 8          # eight ensembles, all four recordings
    assert j["cpu_baseline"]["kind"] in ("reference", "port") and j["cpu_baseline"]["value"] > 0
    if j["cpu_baseline"]["kind"] == "reference":
        # the real reference took EVERY checked ensemble (not only those built from recording 0), its -O3 build and its own PROFILE() marks ran
        assert any("reference backend" in a and str(j["parity_check"]["ensembles"]) in a for a in j["parity_check"]["against"])
        assert j["cpu_baseline"]["stage_ms"] and "ProcessSymbol" in j["cpu_baseline"]["stage_ms"] and "DADeconvolve" in j["cpu_baseline"]["stage_ms"]
        assert j["cpu_baseline"]["o3"] is None or j["cpu_baseline"]["o3"].get("value", 1) > 0
    assert "facade" in j and "host_u8" in j and j["host_u8"].get("x_real_time", 0) > 0
    # the heterogeneous multiplex behind the headline: its own throughput, Viterbi stage time and parity leg (every sub-channel)
    het = j["extras"]["hetero"]
    assert het.get("parity") is True and het["value"] > 0 and het["msc_viterbi_ms"] > 0, het
    # ... and the batch of independent ensembles (five multiplexes, five selections): throughput, Viterbi stage time, its own parity leg
    mix = j["extras"]["mixed_layouts"]
    assert mix.get("parity") is True and mix["value"] > 0 and mix["msc_viterbi_ms"] > 0, mix
    # the headline's geometry on signals as a receiver meets them: drifting sample clocks (the find chain), 6-10 dB; own parity legs
    for kind in ("drift", "low_snr"):
        leg = j["extras"][kind]
        assert leg.get("parity") is True and leg["value"] > 0 and leg["wide_sync_stats"]["frames"] > 0, leg
    assert j["extras"]["drift"]["wide_sync_stats"]["frames_accepted_from_the_wide_pass"] > 0
    # every service's logical frames on the host, every step, through the bulk drain overlapped with the next step
    assert j["msc_drain"]["services"] == 8 * 18 and j["msc_drain"]["value"] > 0 and j["msc_drain"]["logical_frames_per_step"] == 8 * 18 * 40, j["msc_drain"]
    # INTEGRATION level 2 as a build: the reference backend with one file replaced by a seam binding, next to the unmodified build
    l2 = j["facade"].get("level2", {})
    assert all(l2.get(k, {}).get("x_realtime", 0) > 0 for k in ("reference", "l2a", "l2b")), l2
    # the Viterbi seam's shared decoder combines the sub-channels of a CIF: more than one code word per device call with 18 services
    assert l2["l2b"]["all_18_services"]["code_words_per_device_call"] > 1.5 and l2["l2b"]["all_18_services"]["cpu_ms_per_frame"] > 0, l2["l2b"]
    assert j["roofline"]["measured_copy_GBps"] > 1000 and 0 < j["roofline"]["frac_of_achievable"] < 1.2
    assert j["profile_build"]["src_sha256"] and j["profile_build"]["lib_sha256"]


def test_rccl_gather_of_device_buffers_one_rank(gpu):
    j = run_bench(["--gpus", "1", "--steps", "2", "--ensembles", "8", "--frames", "10", "--no-alt-schedule", "--no-cpu-baseline", "--no-extras"], {"DABPHY_FORCE_DIST": "1"})
    assert j["n_gpus"] == 1 and j["rccl_ranks"] == 1 and j["value"] > 0


def test_gpus_2_starts_its_ranks(gpu):
    import torch
    env = {} if torch.cuda.device_count() >= 2 else {"DABPHY_SHARE_GPU": "1", "DABPHY_DIST_BACKEND": "gloo"}
    env = dict(env, **{k: "" for k in ()})
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        os.environ.pop(k, None)
    j = run_bench(["--gpus", "2", "--steps", "1", "--ensembles", "4", "--frames", "10", "--no-alt-schedule"], env)
    assert j["n_gpus"] == 2 and j["config"]["ensembles_per_gpu"] == 4
    assert j["rccl_ranks"] == (2 if torch.cuda.device_count() >= 2 else 0)
    # the multi-rank line is as complete as the single-GPU one: roofline of the slowest rank (per-rank kernel times beside it) and the
    # CPU baseline, timed on rank 0's host cores after the timed region
    assert j["roofline"]["bound"] == "hbm" and 0 < j["roofline"]["frac"] < 1 and len(j["roofline"]["kernel_ms_per_rank"]) == 2
    assert j["roofline"]["kernel_ms"] == max(j["roofline"]["kernel_ms_per_rank"])
    assert j["cpu_baseline"]["kind"] in ("reference", "port") and j["cpu_baseline"]["value"] > 0 and j["cpu_baseline"]["cores"] >= 1
    # every rank proved its own shard against the oracle outside the timed region
    assert j["parity_check"]["ranks_ok"] == 2 and j["parity_check"]["fib_equal"] and j["parity_check"]["msc_equal"], j["parity_check"]
    # (round 5 also compared the baseline with a one-rank run of the same configuration: dropped from the device suite in round 6 -- 20 s for a
    # figure that moves by 10 % with the host's other load)
