#!/bin/bash
# per-kernel times of the headline frame on one lane (rocprofv3 --kernel-trace --stats), development aid.  usage: r03_stats.sh <tag> [spp]
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-x}; SPP=${2:-64}
O=$R/gpurun_out/r03/stats_$TAG; mkdir -p $O
export TMPDIR=/tmp RT_WF_LANES=1
cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/scripts/prof_target.py $SPP 4 4 > $O/run.log 2>&1
f=$(ls -t $O/*/*_kernel_stats.csv | head -1)
python3 - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    n=r["Name"][:90]
    print(f'{float(r["TotalDurationNs"])/1e6/4:9.3f} ms/frame  calls {r["Calls"]:>5}  {n}')
P
tail -1 $O/run.log
