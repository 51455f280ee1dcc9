#!/bin/bash
# Not a test: round 4, closing device session on the round's final sources: the whole -m gpu suite, tools/make_profiles.sh (kernel statistics
# + counter passes), a kernel trace for the step's time line, then the default bench line.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4k; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
tail -n 4 $O/gputest.log
bash tools/make_profiles.sh > $O/make_profiles.log 2>&1
rm -rf gpurun_out/kt; mkdir -p gpurun_out/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -o kt -- python bench.py --no-cpu-baseline --no-alt-schedule --no-extras > $O/kt.log 2>&1
python tools/step_timeline.py $(find gpurun_out/kt -name "kt_kernel_trace.csv" | head -1) > $O/step_timeline.txt 2>&1
cat $O/step_timeline.txt
