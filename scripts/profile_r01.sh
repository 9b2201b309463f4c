#!/bin/bash
# rocprofv3 passes for bench.py's workload (run on the GPU box through gpurun).
# usage: scripts/profile_r01.sh <tag> [bench args...]
set -e
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--no-cpu-baseline $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/bench_stats.json 2> $OUT/stats.err
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
echo "write pass done"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_l2 -- python3 $R/bench.py $ARGS > $OUT/bench_pmc_l2.json 2> $OUT/pmc_l2.err
echo "l2 pass done"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $ARGS > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.err || echo "sq pass failed"
echo "sq pass done"
find $OUT -name "*.csv" | head -50
