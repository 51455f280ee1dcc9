#!/bin/bash
# Not a test: the random parity sweeps on the kernels with the block-parallel traceback: independent ensembles with selection changes, random
# multiplexes (all four decode shapes), random channels through the single-frame path and batch mode (decode_shape 2).  Every line ends in
# "equal"; a mismatch raises.  (This session's sweep of independent ensembles is the one that FOUND the replayed-batch re-acquisition
# difference, seed 2026 trial 33 -- DESIGN.md section 7, exact batch mode; tools/r05_gpu_run11.sh repeats it on the fixed library.)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5m; rm -rf $O; mkdir -p $O
timeout 500 python tools/sweep_independent.py 120 2026 > $O/independent.txt 2>&1; echo "independent rc $?"; tail -1 $O/independent.txt
timeout 300 python tools/sweep_multiplex.py 60 77 > $O/multiplex.txt 2>&1; echo "multiplex rc $?"; tail -1 $O/multiplex.txt
timeout 200 python tools/sweep_correctors.py 40 5 > $O/correctors.txt 2>&1; echo "correctors rc $?"; tail -1 $O/correctors.txt
SWEEP_DECODE_SHAPE=2 timeout 200 python tools/sweep_batch.py 30 9 > $O/batch_shape2.txt 2>&1; echo "batch rc $?"; tail -1 $O/batch_shape2.txt
