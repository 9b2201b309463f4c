#!/bin/bash
for ct in 0.4 0.55 0.7 0.85 1.0; do echo "== cost_traverse=$ct"; RT_BVH_COST_TRAVERSE=$ct python scripts/prof_target.py 64 4 3; RT_BVH_COST_TRAVERSE=$ct python scripts/bistro_perf.py | head -1; done
