#!/bin/bash
# frame time vs samples per batch of the wavefront pipeline (development aid)
for b in 3 4 6 8 11 16 22 32; do
  echo "== RT_WF_BATCH=$b"
  RT_WF_BATCH=$b python scripts/prof_target.py ${1:-64} 4 2
done
