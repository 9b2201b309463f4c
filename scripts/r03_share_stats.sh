#!/bin/bash
# per-kernel times of rank 0's eighth of the headline frame (rocprofv3 --kernel-trace --stats), development aid.  usage: r03_share_stats.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-x}
O=$R/gpurun_out/r03/sharestats_$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/scripts/share_target.py 8 6 > $O/run.log 2>&1
f=$(ls -t $O/*/*_kernel_stats.csv | head -1)
python3 - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=0
for r in rows[:12]:
    n=r["Name"][:80]
    print(f'{float(r["TotalDurationNs"])/1e6/6:9.3f} ms/frame  calls {r["Calls"]:>5}  avg {float(r["AverageNs"])/1e3:8.1f} us  {n}')
P
tail -1 $O/run.log
