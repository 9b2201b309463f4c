#!/bin/bash
# frame time and node visits of the headline frame with trees from the different builders
# (RT_BUILD_METHOD: 0 host binned SAH + reinsertion, 1 host PLOC, 2 device PLOC = default)
run() { echo "== $1"; shift; env "$@" python scripts/prof_target.py 64 4 3; env "$@" python scripts/wf_diag.py 4 | tail -2 | head -1; }
run "host sah + reinsertion" RT_BUILD_METHOD=0
run "host ploc r=16" RT_BUILD_METHOD=1 RT_PLOC_RADIUS=16
run "device ploc r=8" RT_BUILD_METHOD=2 RT_PLOC_RADIUS=8
run "device ploc r=16 (default)" RT_BUILD_METHOD=2 RT_PLOC_RADIUS=16
run "device ploc r=32" RT_BUILD_METHOD=2 RT_PLOC_RADIUS=32
