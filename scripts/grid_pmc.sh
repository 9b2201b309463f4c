#!/bin/bash
# PMC passes of the light-grid shadow stage on scripts/grid_prof.py (run on the GPU box through gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_gridpmc
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
T="timeout -k 10 240"
pass() { name=$1; shift; $T rocprofv3 "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $R/scripts/grid_prof.py ${GRID_ARGS} > $OUT/$name.out 2> $OUT/$name.err && echo "$name done" || { echo "$name FAILED"; tail -3 $OUT/$name.err; return 1; }; }
pass sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS || exit 1
pass fetch --pmc FETCH_SIZE || exit 1
pass l2 --pmc TCC_HIT_sum TCC_MISS_sum || exit 1
pass sq2 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM || exit 1
cd $R
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); dur = collections.Counter()
for f in glob.glob('gpurun_out/prof_gridpmc/*/*/*counter_collection.csv'):
    seen=set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-40:]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for f in glob.glob('gpurun_out/prof_gridpmc/sq/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-40:]; calls[k]+=1; dur[k]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for k in agg:
    if 'k_wf_' not in k: continue
    a = agg[k]; ms = dur[k]/1e6
    print(k, 'calls', calls[k], 'ms', round(ms,2))
    if a.get('SQ_WAVE_CYCLES'): print('   wait_any/wave_cycles', round(a['SQ_WAIT_ANY']/a['SQ_WAVE_CYCLES'],3), 'valu_busy', round(4*a['SQ_ACTIVE_INST_VALU']/(ms*1e-3*2.4e9*1024),3), 'VALU insts', a['SQ_INSTS_VALU'], 'VMEM_RD', a['SQ_INSTS_VMEM_RD'], 'waves', a['SQ_WAVES'])
    if a.get('FETCH_SIZE'): print('   fetch GB', round(a['FETCH_SIZE']*64/1e9*2,2), '(x2 corrected)', 'L2 hit', round(a['TCC_HIT_sum']/max(1,a['TCC_HIT_sum']+a['TCC_MISS_sum']),3), 'L2 req', a['TCC_HIT_sum']+a['TCC_MISS_sum'])
    print('   ', {c: v for c, v in a.items() if c.startswith('SQ_') and c not in ('SQ_WAVE_CYCLES',)})
PY
