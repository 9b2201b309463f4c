#!/bin/bash
# A/B on one box: alternate the default library and every build/variants/*.so, <reps> times (default 3); each run renders
# the headline frame 5 times and prints the last frame's kernel ms and Mrays/s
reps=${1:-3}
for rep in $(seq $reps); do
echo "== default"; python scripts/prof_target.py 64 4 5
for f in build/variants/*.so; do echo "== $f"; RT_HIP_LIB=$PWD/$f python scripts/prof_target.py 64 4 5; done
done
