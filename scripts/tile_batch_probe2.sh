#!/bin/bash
for b in 22 32 64; do echo "== RT_WF_BATCH=$b"; RT_WF_BATCH=$b python scripts/tile_probe2.py 2>&1 | grep -E "tile (128|64):"; done
