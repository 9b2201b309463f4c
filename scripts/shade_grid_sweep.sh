#!/bin/bash
# grid size of the generate / shade / finish kernels (256-thread blocks per CU), headline frame
for rep in 1 2; do
for b in ${BLOCKS:-8 4 5 10 16 20}; do echo "== $b blocks per CU"; RT_WF_SHADE_BLOCKS_PER_CU=$b python scripts/prof_target.py 64 4 4; done
done
