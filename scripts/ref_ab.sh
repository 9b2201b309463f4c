#!/bin/bash
# reference-mode kernel (mode 1, sponza-like 1080p and bistro-like 4K): default library vs build/variants/*.so
for rep in 1 2; do
for lib in default build/variants/*.so; do
  echo "== $lib"
  if [ $lib = default ]; then python bench.py --mode reference --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], d['ms_per_step'])"
  else RT_HIP_LIB=$PWD/$lib python bench.py --mode reference --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_avg_ms'], d['ms_per_step'])"; fi
done; done
