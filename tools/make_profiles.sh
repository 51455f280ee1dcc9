#!/bin/bash
# Not a test: collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun_out/prof/stats_*          --kernel-trace --stats of the default bench command
#   gpurun_out/prof/pmc_fetch_*      --pmc FETCH_SIZE   (own pass)
#   gpurun_out/prof/pmc_write_*      --pmc WRITE_SIZE   (own pass)
#   gpurun_out/prof/pmc_vit_*        SQ instruction / LDS counters of k_viterbi and k_msc_gather (own pass)
#   gpurun_out/prof/pmc_sq{1,2}_*    SQ issue / wait / LDS counters of k_demod (own passes, demod-only driver)
#   gpurun_out/prof/drift_*          --kernel-trace --stats of bench.py's drift leg alone (tools/bench_channel.py)
#   gpurun_out/prof/valu_rate.txt    tools/ubench/valu_rate.hip: issue cost of the instructions the Viterbi kernel is made of
# The summaries are then copied into profiles/ by tools/collect_profiles.py <tag>.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/prof; rm -rf $O; mkdir -p $O
BENCH="python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-alt-schedule --no-extras $BENCH_EXTRA"
# (the statistics pass runs bench.py's own default step counts -- 10 after 4 warm-up steps --, so that its per-kernel averages are dominated by
# the steps the bench line times; the counter passes stay short)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats -- python bench.py --no-cpu-baseline --no-alt-schedule --no-extras $BENCH_EXTRA > $O/stats.log 2>&1
KR='k_demod|k_viterbi|k_msc_gather|k_sync_|k_fic_gather|k_rs_msc|k_superframe|k_acquire|k_snr'
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$KR" --output-format csv -d $O -o pmc_fetch -- $BENCH > $O/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$KR" --output-format csv -d $O -o pmc_write -- $BENCH > $O/pmc_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-include-regex 'k_viterbi|k_msc_gather' --output-format csv -d $O -o pmc_vit -- $BENCH > $O/pmc_vit.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --kernel-include-regex k_demod --output-format csv -d $O -o pmc_sq1 -- python tools/prof_demod.py > $O/pmc_sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE --kernel-include-regex k_demod --output-format csv -d $O -o pmc_sq2 -- python tools/prof_demod.py > $O/pmc_sq2.log 2>&1
hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate && timeout 120 /tmp/valu_rate > $O/valu_rate.txt 2>&1
# the drift leg (every ensemble its own sampling-clock offset: the find chain) under the kernel tracer, on its own
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o drift -- python tools/bench_channel.py drift > $O/drift.log 2>&1
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err
# (gpurun copies at most 64 MiB back: the per-dispatch traces of the long passes stay on the box, the statistics are what is kept)
find $O -name "*kernel_trace.csv" -size +4M -delete; find $O -name "*.db" -delete; find $O -name "*.pftrace" -delete
tail -n 2 $O/*.log | cut -c1-300
du -sh $O gpurun_out
find $O -name "*.csv" | xargs ls -la
