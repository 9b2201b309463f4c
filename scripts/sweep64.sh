#!/bin/bash
# headline frame (64 spp, 4 bounces) with the default library and every variant under build/variants
echo "== default"; python scripts/prof_target.py 64 4 3
for f in build/variants/*.so; do echo "== $f"; RT_HIP_LIB=$PWD/$f python scripts/prof_target.py 64 4 3; done
