#!/bin/bash
# 8-wide experiment A/B: 4-wide default, 8-wide default library, and every build/variants/*.so with RT_BVH8=1
for rep in 1 2; do
echo "== 4-wide"; python scripts/prof_target.py 64 4 4
echo "== 8-wide default"; RT_BVH8=1 python scripts/prof_target.py 64 4 4
for f in build/variants/*.so; do echo "== 8-wide $f"; RT_BVH8=1 RT_HIP_LIB=$PWD/$f python scripts/prof_target.py 64 4 4; done
done
