# Not a test: occupancy / issue counters of k_demod alone for a list of library builds (run through gpurun from the repo root)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/occ; rm -rf $O; mkdir -p $O
for L in "$@"; do
  DABPHY_LIB=$PWD/welle.io_amd/$L.so PROBE_CHUNK=25 PROBE_F=20 timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --kernel-include-regex k_demod --output-format csv -d $O -o $L -- python tools/prof_demod.py > $O/$L.log 2>&1
done
