#!/bin/bash
# run scripts/ab_extended.py for every kernel-variant library under build/variants
for f in build/variants/*.so; do
  echo "== $f"
  RT_HIP_LIB=$PWD/$f python scripts/ab_extended.py ${1:-8} ${2:-wf} 2>&1 | grep -E "spp 4b|0b|exact"
done
