#!/bin/bash
# like sweep64.sh but alternates default and variants twice (box-to-box and run-to-run noise is ~1.5 %)
for rep in 1 2; do
echo "== default"; python scripts/prof_target.py 64 4 3
for f in build/variants/*.so; do echo "== $f"; RT_HIP_LIB=$PWD/$f python scripts/prof_target.py 64 4 3; done
done
