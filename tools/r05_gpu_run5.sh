#!/bin/bash
# Not a test: kernel timelines of latency-mode calls (what is a medium-batch call made of besides the decode launch?)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5i; rm -rf $O; mkdir -p $O
for g in "32 16 0" "32 16 1" "1 1 0" "16 8 0"; do
  set -- $g; t=${1}x${2}_s${3}
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O -o kt_$t -- python tools/trace_call.py $1 $2 $3 > $O/$t.log 2>&1
  f=$(find $O -name "kt_${t}_kernel_trace.csv" | head -1)
  echo "== $t: $(grep 'ms per call' $O/$t.log)" >> $O/timelines.txt
  python tools/step_timeline.py $f >> $O/timelines.txt 2>&1
done
cat $O/timelines.txt
