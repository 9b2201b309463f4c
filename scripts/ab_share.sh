#!/bin/bash
# like ab.sh for rank 0's eighth of the headline frame
reps=${1:-3}
for rep in $(seq $reps); do
echo "== default"; python scripts/share_target.py 8 6 | tail -1
for f in build/variants/*.so; do echo "== $f"; RT_HIP_LIB=$PWD/$f python scripts/share_target.py 8 6 | tail -1; done
done
