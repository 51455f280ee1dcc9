#!/bin/bash
# Not a test: the block-parallel traceback (k_traceback_sp2, a wave per stretch of the code word) on the device:
#  1. the state-parallel device tests with the experiments build, the traceback pass forced for every size and every guess forced WRONG
#     (DABPHY_SP2_TB_WARM=0: each stretch is walked again from the true state, the rounds cascade) and with 1 block of run-in;
#  2. the same tests with the product library;  3. the three decoders against each other over batch sizes;  4. kernel timelines.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5j; rm -rf $O; mkdir -p $O
K="state_parallel or shallow or two_kernel or mixed or decode_shape or sp2 or pair_exchange"
for warm in 0 1; do
  DABPHY_LIB=$PWD/gpurun_in/lib_exp.so DABPHY_SP2_TB_MIN_CW=0 DABPHY_SP2_TB_WARM=$warm timeout 500 python -m pytest tests/test_gpu_stream.py tests/test_gpu_parity.py -m gpu -x -q -k "$K" > $O/tests_warm$warm.log 2>&1
  echo "warm $warm: $(tail -1 $O/tests_warm$warm.log)"
done
timeout 500 python -m pytest tests/test_gpu_stream.py tests/test_gpu_parity.py tests/test_gpu_bench_config.py -m gpu -x -q -k "$K" > $O/tests_product.log 2>&1; echo "product: $(tail -1 $O/tests_product.log)"
timeout 600 python tools/sweep_decode_shape.py > $O/sweep.txt 2>&1; cat $O/sweep.txt
# the traceback pass forced at every size (experiments build): where does it start to pay?
DABPHY_LIB=$PWD/gpurun_in/lib_exp.so DABPHY_SP2_TB_MIN_CW=0 timeout 600 python tools/sweep_decode_shape.py > $O/sweep_split_always.txt 2>&1; cat $O/sweep_split_always.txt
for g in "32 16 0" "16 8 0"; do
  set -- $g; t=${1}x${2}_s${3}
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O -o kt_$t -- python tools/trace_call.py $1 $2 $3 > $O/$t.log 2>&1
  f=$(find $O -name "kt_${t}_kernel_trace.csv" | head -1)
  echo "== $t: $(grep 'ms per call' $O/$t.log)" >> $O/timelines.txt
  python tools/step_timeline.py $f >> $O/timelines.txt 2>&1
done
cat $O/timelines.txt
