#!/bin/bash
echo "== default"; python scripts/rank_share.py
for f in build/variants/*.so; do echo "== $f"; RT_HIP_LIB=$PWD/$f python scripts/rank_share.py; done
