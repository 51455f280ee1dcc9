#!/bin/bash
# Not a test: the random sweeps that reach exact batch mode's second pass, on the library with the history ring put back: independent
# ensembles with selection changes (the seed that found the difference, and a fresh one), low-SNR batch streams in exact mode.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5o; rm -rf $O; mkdir -p $O
timeout 400 python tools/sweep_independent.py 120 2026 > $O/independent_2026.txt 2>&1; echo "independent 2026 rc $?"; grep -v ": equal" $O/independent_2026.txt | tail -4
timeout 400 python tools/sweep_independent.py 80 31337 > $O/independent_31337.txt 2>&1; echo "independent 31337 rc $?"; grep -v ": equal" $O/independent_31337.txt | tail -4
timeout 300 python tools/sweep_batch.py 40 11 exact > $O/batch_exact.txt 2>&1; echo "batch exact rc $?"; tail -1 $O/batch_exact.txt
