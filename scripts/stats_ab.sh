#!/bin/bash
# per-kernel times (rocprofv3 --kernel-trace --stats) of the headline frame for the default library and every build/variants/*.so
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
run() { tag=$1; lib=$2; RT_HIP_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_ab/$tag -- python3 $R/scripts/prof_target.py ${SPP:-64} ${BOUNCES:-4} 4 > $R/gpurun_out/stats_ab/$tag.out 2>&1
  python3 - $R/gpurun_out/stats_ab/$tag <<PY
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
print("==", sys.argv[1].split("/")[-1])
for r in csv.DictReader(open(f)):
    n = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", r["Name"])
    if n: print(f"  {n.group(0):28s} calls {int(r['Calls']):4d}  total {float(r['TotalDurationNs'])/1e6/4:8.2f} ms/frame  {r['Percentage']}%")
PY
}
mkdir -p $R/gpurun_out/stats_ab
run default $R/gpu_raytracer_amd/librt_hip.so
for f in $R/build/variants/*.so; do run $(basename $f .so) $f; done
