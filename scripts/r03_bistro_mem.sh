# round 3: memory of the bistro-like scene's light grids against frame time (sorted prefix / walk limit, `heavy`), development aid
run() { python bench.py --scene bistro_like --spp 16 --bounces 4 --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); g=d['roofline'].get('shadow_grids') or {}; print(round(d['ms_per_step'],2), 'ms', round(g.get('bytes',0)/1e9,2), 'GB answered', round(g.get('shadow_segments_answered_share',0),3), 'entries/answered', round(g.get('entries_read_per_answered_segment',0),2))"; }
echo "== prefix 32 heavy 128"; run
echo "== prefix 32 heavy 96"; RT_SHADOW_GRID_HEAVY=96 run
echo "== prefix 32 heavy 64"; RT_SHADOW_GRID_HEAVY=64 run
export RT_HIP_LIB=$PWD/build/variants/p24.so
echo "== prefix 24 heavy 128"; run
echo "== prefix 24 heavy 96"; RT_SHADOW_GRID_HEAVY=96 run
echo "== prefix 24 heavy 64"; RT_SHADOW_GRID_HEAVY=64 run
