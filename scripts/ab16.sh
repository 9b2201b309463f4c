#!/bin/bash
# like ab.sh at 16 spp, with the per-kernel split left to rocprof runs; prints kernel ms of the last of 4 frames
reps=${1:-2}
for rep in $(seq $reps); do
echo "== default"; python scripts/prof_target.py 16 4 4
for f in build/variants/*.so; do echo "== $f"; RT_HIP_LIB=$PWD/$f python scripts/prof_target.py 16 4 4; done
done
