#!/bin/bash
# rocprofv3 passes for one bench.py workload (run on the GPU box through gpurun): kernel stats, then PMC passes each in
# its own run (FETCH_SIZE and WRITE_SIZE do not fit one pass; SQ has 8 slots), never combined with other trace domains.
# The program itself follows `--` (python3 bench.py ...).  Every pass is bounded by its own timeout; TA_* / TCP_*
# counters are not collected (they hang rocprofv3 on this pool).
# usage: scripts/profile_r03.sh <tag> [bench args...]      then: python scripts/summarize_profile.py gpurun_out/prof_<tag> profiles/<name> ...
TAG=${1:-r03}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r03_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# one lane: on two lanes (the default with light grids) the other lane's kernels run beside the one being timed and every duration in
# the trace includes the waiting for them; bench.py reports the one-lane frame time next to the two-lane one for the comparison
export RT_WF_LANES=${RT_WF_LANES:-1}
cd /tmp
ARGS="--no-cpu-baseline $@"
T="timeout -k 10 ${PROFILE_TIMEOUT:-300}"
pass() { name=$1; shift; $T rocprofv3 "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $R/bench.py $ARGS > $OUT/bench_$name.json 2> $OUT/$name.err && echo "$name pass done" || { echo "$name pass FAILED"; tail -3 $OUT/$name.err; return 1; }; }
pass stats --stats || exit 1
pass pmc_fetch --pmc FETCH_SIZE || exit 1
pass pmc_write --pmc WRITE_SIZE || exit 1
pass pmc_l2 --pmc TCC_HIT_sum TCC_MISS_sum || exit 1
pass pmc_sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS || exit 1
# issue side: what the instruction issue slots of the traversal kernels are spent on
pass pmc_sq2 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU || exit 1
pass pmc_sq3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC || echo "sq3 optional"
