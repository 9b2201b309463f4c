#!/bin/bash
# usage: stats_target.sh <tag> <spp> <bounces> [v1|sm]
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/stats_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -- python3 $R/scripts/prof_target.py $2 $3 2 $4 > $OUT/out.txt 2> $OUT/err.txt
cat $OUT/s/*/*_kernel_stats.csv | cut -c1-220
cat $OUT/out.txt
