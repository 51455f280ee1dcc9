#!/bin/bash
# Not a test: the block-parallel traceback as the product has it (k_viterbi_sp2 always followed by k_traceback_sp2; four stretches per code
# word, three for launches of more than 512 groups) on the device:
#  1. the state-parallel device tests with the experiments build and every guess forced WRONG (DABPHY_SP2_TB_WARM=0: each stretch is walked
#     again from the true state, the rounds cascade), four and three waves per work-group; with one block of run-in;
#  2. the same tests with the product library;  3. the three decoders against each other over batch sizes;  4. kernel timelines.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5k; rm -rf $O; mkdir -p $O
K="state_parallel or shallow or two_kernel or mixed or decode_shape or sp2 or pair_exchange or either_decoder or lane_exchanges"
X=$PWD/gpurun_in/lib_exp.so
DABPHY_LIB=$X DABPHY_SP2_TB_WARM=0 timeout 500 python -m pytest tests/test_gpu_stream.py tests/test_gpu_parity.py -m gpu -x -q -k "$K" > $O/tests_warm0.log 2>&1; echo "warm 0: $(tail -1 $O/tests_warm0.log)"
DABPHY_LIB=$X DABPHY_SP2_TB_WARM=0 DABPHY_SP2_TB_RESIDENT=0 timeout 500 python -m pytest tests/test_gpu_stream.py tests/test_gpu_parity.py -m gpu -x -q -k "$K" > $O/tests_warm0_three.log 2>&1; echo "warm 0, three waves: $(tail -1 $O/tests_warm0_three.log)"
DABPHY_LIB=$X DABPHY_SP2_TB_WARM=1 timeout 500 python -m pytest tests/test_gpu_stream.py tests/test_gpu_parity.py -m gpu -x -q -k "$K" > $O/tests_warm1.log 2>&1; echo "warm 1: $(tail -1 $O/tests_warm1.log)"
DABPHY_LIB=$X DABPHY_SP2_TB_RESIDENT=0 timeout 500 python -m pytest tests/test_gpu_stream.py tests/test_gpu_parity.py -m gpu -x -q -k "$K" > $O/tests_three.log 2>&1; echo "three waves: $(tail -1 $O/tests_three.log)"
timeout 500 python -m pytest tests/test_gpu_stream.py tests/test_gpu_parity.py tests/test_gpu_bench_config.py -m gpu -x -q -k "$K" > $O/tests_product.log 2>&1; echo "product: $(tail -1 $O/tests_product.log)"
timeout 600 python tools/sweep_decode_shape.py > $O/sweep.txt 2>&1; cat $O/sweep.txt
for g in "32 16 0" "16 8 0" "4 4 0" "1 16 0"; do
  set -- $g; t=${1}x${2}_s${3}
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O -o kt_$t -- python tools/trace_call.py $1 $2 $3 > $O/$t.log 2>&1
  f=$(find $O -name "kt_${t}_kernel_trace.csv" | head -1)
  echo "== $t: $(grep 'ms per call' $O/$t.log)" >> $O/timelines.txt
  python tools/step_timeline.py $f >> $O/timelines.txt 2>&1
done
cat $O/timelines.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
j=json.load(open("gpurun_out/r5k/bench.json"))
print(j["value"], j["ms_per_step"], j["roofline"]["frac"])
for k,v in j.get("extras",{}).items(): print(k, json.dumps(v)[:600])
PY
