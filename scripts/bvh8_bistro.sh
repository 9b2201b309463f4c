#!/bin/bash
# 8-wide experiment on the bistro-like scene (deep tree) and upload cost
for rep in 1 2; do
echo "== bistro 4-wide"; python bench.py --scene bistro_like --spp 16 --bounces 4 --steps 3 --warmup 1 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['nodes_per_ray'], d['roofline']['tris_per_ray'])"
echo "== bistro 8-wide"; RT_BVH8=1 python bench.py --scene bistro_like --spp 16 --bounces 4 --steps 3 --warmup 1 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['nodes_per_ray'], d['roofline']['tris_per_ray'])"
done
echo "== upload 4"; python scripts/upload_time.py
echo "== upload 8"; RT_BVH8=1 python scripts/upload_time.py
for ml in 2 3; do echo "== sponza 8-wide max_leaf $ml"; RT_BVH8=1 RT_BVH_MAX_LEAF=$ml python scripts/prof_target.py 64 4 4; done
