#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/share_trace; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/scripts/share_trace.py > $OUT/out.txt 2> $OUT/err.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/t/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_wf" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 3
last = rows[-n:]
span = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e6
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) / 1e6
gaps = sorted(((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(last, last[1:])), reverse=True)
print(f"{n} kernels per frame: span {span:.2f} ms, sum of kernels {busy:.2f} ms, gaps {span-busy:.2f} ms (largest us: {[round(g,1) for g in gaps[:6]]})")
import collections
d = collections.defaultdict(float)
for r in last:
    k = r["Kernel_Name"]; k = ("trace_any" if "<false, true>" in k else "trace_closest") if "k_wf_trace" in k else k.split("::")[-1].split("(")[0]
    d[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print({k: round(v, 2) for k, v in d.items()})
PY
cat $OUT/out.txt
