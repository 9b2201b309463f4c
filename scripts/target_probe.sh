#!/bin/bash
# headline frame vs paths in flight per batch (development aid)
for t in 67108864 100000000 134000000; do for rep in 1 2; do echo "== RT_WF_TARGET_PATHS=$t"; RT_WF_TARGET_PATHS=$t python scripts/prof_target.py 64 4 3; done; done
