# round 3: the bistro-like 4K share (64 spp, 8 bounces, an eighth of the tiles) on one and two lanes by samples per batch (development aid)
run() { python bench.py --scene bistro_like --width 3840 --height 2160 --spp 64 --bounces 8 --share-of 8 --steps 2 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); f=d['roofline']['frame']; print('two lanes', round(f['kernel_avg_ms'],1), 'one lane', round(f['kernel_avg_ms_one_lane'],1))"; }
echo "== default"; run
for b in 8 16 32; do echo "== RT_WF_BATCH=$b"; RT_WF_BATCH=$b run; done
