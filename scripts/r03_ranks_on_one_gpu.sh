#!/bin/bash
# how much concurrency the chip can use: the headline frame by 1, 2, 3, 4 processes sharing ONE GPU (gloo), development aid
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r03
for n in 1 2 4 1 2 3; do
  if [ $n = 1 ]; then timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03/ranks_$n.json 2>/dev/null
  else RT_BENCH_BACKEND=gloo RT_BENCH_DEVICE=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2954$n bench.py --gpus $n --steps 6 --warmup 2 > gpurun_out/r03/ranks_$n.json 2>/dev/null; fi
  python - <<P
import json
l=[x for x in open("gpurun_out/r03/ranks_$n.json").read().splitlines() if x.startswith("{")][-1]; p=json.loads(l)
print("$n process(es):", round(p["ms_per_step"],2), "ms", round(p["value"]), "Mrays/s crc", p["config"]["frame_crc"])
P
done
