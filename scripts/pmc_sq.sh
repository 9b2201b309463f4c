#!/bin/bash
# usage: pmc_sq.sh <tag> <spp> <bounces> [v1]   — SQ counters only (TA_* counter passes hang on this pool)
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="$2 $3 2 $4"
run() { name=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $R/scripts/prof_target.py $ARGS > $OUT/$name.out 2> $OUT/$name.err || echo "$name failed"; }
run sq GRBM_GUI_ACTIVE GRBM_TA_BUSY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
run sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_WAVES SQ_ACTIVE_INST_ANY
python3 - <<PY
import csv, glob, collections
for name in ["sq","sq2"]:
    fs = glob.glob("$OUT/%s/*/*_counter_collection.csv" % name)
    if not fs: print(name, "no data"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "k_render_" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): print(name, k, "%.5g" % (sum(v)/len(v)))
PY
cat $OUT/sq.out
