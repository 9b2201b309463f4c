"""Per-kernel summary of the PMC passes written by scripts/grid_pmc.sh (development aid)."""
import csv, glob, collections, re, sys
root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof_gridpmc'
def kname(n):
    m = re.search(r'(k_[a-z_0-9]+(<[^>]*>)?)', n)
    return m.group(1) if m else n[:30]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); dur = collections.Counter()
for f in glob.glob(root + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[kname(r['Kernel_Name'])][r['Counter_Name']] += float(r['Counter_Value'])
for f in glob.glob(root + '/sq/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        k = kname(r['Kernel_Name']); calls[k] += 1; dur[k] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for k in sorted(agg, key=lambda k: -dur[k]):
    if 'k_wf_' not in k: continue
    a = agg[k]; ms = dur[k] / 1e6
    print(k, 'calls', calls[k], 'ms', round(ms, 2))
    if a.get('SQ_WAVE_CYCLES'):
        print('   wait_any/wave_cycles', round(a['SQ_WAIT_ANY'] / a['SQ_WAVE_CYCLES'], 3), 'valu_busy', round(4 * a['SQ_ACTIVE_INST_VALU'] / (ms * 1e-3 * 2.4e9 * 1024), 3),
              'VALU insts', a['SQ_INSTS_VALU'], 'VMEM_RD', a['SQ_INSTS_VMEM_RD'], 'waves', a['SQ_WAVES'], 'waves/SIMD', round(a['SQ_WAVE_CYCLES'] / max(1, a['SQ_BUSY_CYCLES']) / 4, 2))
    if a.get('FETCH_SIZE'):
        print('   fetch GB', round(a['FETCH_SIZE'] * 64 / 1e9 * 2, 2), '(x2 corrected)', 'L2 hit', round(a['TCC_HIT_sum'] / max(1, a['TCC_HIT_sum'] + a['TCC_MISS_sum']), 3), 'L2 req', a['TCC_HIT_sum'] + a['TCC_MISS_sum'])
    print('   ', {c: v for c, v in a.items() if c.startswith('SQ_')})
