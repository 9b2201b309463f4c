#!/bin/bash
# usage: pmc_passes.sh <tag> <spp> <bounces> [v1]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $R/scripts/prof_target.py $ARGS > $OUT/$name.out 2> $OUT/$name.err || echo "$name failed"; }
ARGS="$@ "
ARGS="${ARGS/ / } 2"
ARGS="$1 $2 2 $3"
run sq GRBM_GUI_ACTIVE GRBM_TA_BUSY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
run sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_WAVES
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_GATE_EN1_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
python3 - <<PY
import csv, glob, collections
for name in ["sq","sq2","ta","tcp","tcp2"]:
    fs = glob.glob("$OUT/%s/*/*_counter_collection.csv" % name)
    if not fs: print(name, "no data"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "k_render_extended" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): print(name, k, "%.4g" % (sum(v)/len(v)))
PY
cat $OUT/sq.out
