#!/bin/bash
# Not a test: round 5, third device session: k_viterbi_sp2 on the device (self-test of its lane exchanges, every state-parallel parity
# test), the decode-shape sweep with the three kernels, the facade's latency, the superframe filter's new verdict kernel in the step.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_stream.py tests/test_gpu_parity.py tests/test_gpu_host_mirror.py tests/test_gpu_impairments.py -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
tail -n 8 $O/gputest.log
timeout 600 python tools/sweep_decode_shape.py > $O/sweep.txt 2>&1; cat $O/sweep.txt
timeout 300 python tools/bench_facade.py > $O/facade.txt 2>&1; tail -3 $O/facade.txt
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-alt-schedule --no-extras --steps 20 > $O/bench_$i.json 2>> $O/bench.err; done
rm -rf gpurun_out/kt; mkdir -p gpurun_out/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -o kt -- python bench.py --no-cpu-baseline --no-alt-schedule --no-extras > $O/kt.log 2>&1
python tools/step_timeline.py $(find gpurun_out/kt -name "kt_kernel_trace.csv" | head -1) > $O/step_timeline.txt 2>&1
cat $O/step_timeline.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5c/bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"], j["stages_ms"])
    except Exception as e:
        print(f, "failed", e)
PY
