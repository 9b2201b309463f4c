#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / L2 / SQ of the light-grid stage for the library in RT_HIP_LIB (or the default), development aid.  usage: r03_gridpmc.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-x}
OUT=$R/gpurun_out/r03/gridpmc_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RT_WF_LANES=1
cd /tmp
T="timeout -k 10 200"
pass() { name=$1; shift; $T rocprofv3 "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $R/scripts/grid_prof.py sponza_like 16 > $OUT/$name.out 2> $OUT/$name.err && echo "$name done" || { echo "$name FAILED"; tail -3 $OUT/$name.err; return 1; }; }
pass sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS || exit 1
pass fetch --pmc FETCH_SIZE || exit 1
pass write --pmc WRITE_SIZE || exit 1
pass l2 --pmc TCC_HIT_sum TCC_MISS_sum || exit 1
cd $R && python3 scripts/grid_pmc_summary.py $OUT | grep -A3 "k_wf_shadow_grid"
