#!/bin/bash
# usage: pmc_sq.sh <tag> <spp> <bounces> [v1|sm]   — SQ counters only (TA_* counter passes hang on this pool)
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="$2 $3 2 $4"
run() { name=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $R/scripts/prof_target.py $ARGS > $OUT/$name.out 2> $OUT/$name.err || echo "$name failed"; }
run sq GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU
run sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAVES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for name in ["sq","sq2"]:
    fs = glob.glob("$OUT/%s/*/*_counter_collection.csv" % name)
    if not fs: print(name, "no data"); continue
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "k_" not in k: continue
        k = k.split("(")[0].replace("void (anonymous namespace)::","").replace("(anonymous namespace)::","")
        if "k_wf_trace" in r["Kernel_Name"]: k = "trace_any" if "<false, true>" in r["Kernel_Name"] else "trace_closest"
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in agg.items():
    wc = c.get("SQ_WAVE_CYCLES", 1)
    print(f"{k:24s} valu_insts={c.get('SQ_INSTS_VALU',0):.3g} lane_util={c.get('SQ_THREAD_CYCLES_VALU',0)/max(1,c.get('SQ_ACTIVE_INST_VALU',1)*64):.2f} wait={c.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst={c.get('SQ_WAIT_INST_ANY',0)/wc:.2f} active={c.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} salu={c.get('SQ_INSTS_SALU',0):.3g} vmem_rd={c.get('SQ_INSTS_VMEM_RD',0):.3g} lds={c.get('SQ_INSTS_LDS',0):.3g} gui={c.get('GRBM_GUI_ACTIVE',0):.3g}")
PY
cat $OUT/sq.out
