#!/bin/bash
echo "== default"; python scripts/bistro_perf.py
for f in build/variants/*.so; do echo "== $f"; RT_HIP_LIB=$PWD/$f python scripts/bistro_perf.py; done
