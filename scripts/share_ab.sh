#!/bin/bash
# A/B of rank 0's share of the headline frame (world 8 and 4) and of the whole frame: default library vs build/variants/*.so
for rep in 1 2; do
for w in 8 4; do
echo "== default share 1/$w"; python scripts/share_target.py $w 4
for f in build/variants/*.so; do echo "== $f share 1/$w"; RT_HIP_LIB=$PWD/$f python scripts/share_target.py $w 4; done
done
echo "== default full"; python scripts/prof_target.py 64 4 4
for f in build/variants/*.so; do echo "== $f full"; RT_HIP_LIB=$PWD/$f python scripts/prof_target.py 64 4 4; done
done
