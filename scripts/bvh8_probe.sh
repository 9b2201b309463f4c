#!/bin/bash
# 8-wide experiment: headline frame, 4-wide default against RT_BVH8=1 at several traversal costs; then counters
for rep in 1 2; do
echo "== 4-wide"; python scripts/prof_target.py 64 4 4
for c in ${COSTS:-0.7 1.0 1.5}; do echo "== 8-wide cost $c"; RT_BVH8=1 RT_BVH8_COST_TRAVERSE=$c python scripts/prof_target.py 64 4 4; done
done
echo "== counters 4-wide"; python scripts/wf_diag.py 4
echo "== counters 8-wide"; RT_BVH8=1 python scripts/wf_diag.py 4
