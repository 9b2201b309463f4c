#!/bin/bash
for ct in 0.6 0.75 0.9 1.0; do echo "== cost_traverse=$ct"; RT_BVH_COST_TRAVERSE=$ct python scripts/bistro_perf.py; RT_BVH_COST_TRAVERSE=$ct python scripts/prof_target.py 64 4 3; done
