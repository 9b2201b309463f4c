#!/bin/bash
# SAH parameter sweep on the headline workload (development aid): RT_BVH_COST_TRAVERSE x RT_BVH_MAX_LEAF
for leaf in 4 3 2; do for ct in 0.25 0.5 1.0 2.0 4.0; do
  echo "== max_leaf=$leaf cost_traverse=$ct"
  RT_BVH_MAX_LEAF=$leaf RT_BVH_COST_TRAVERSE=$ct python scripts/ab_extended.py ${1:-8} wf 2>&1 | grep -E "4b|counters" | tail -2
done; done
