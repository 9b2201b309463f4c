#!/bin/bash
for ct in 0.5 0.75 1.0 1.25; do echo "== cost_traverse=$ct"; RT_BVH_COST_TRAVERSE=$ct python scripts/prof_target.py 64 4 3; done
