# round 3: walk limit 31 (sorted prefix 32) and the `heavy` bound of the light grids, whole frame and an eighth (development aid)
for rep in 1 2 3; do
echo "== eighth default"; python scripts/share_target.py 8 6 | tail -1
for h in 64 128 256; do echo "== eighth walk31 heavy $h"; RT_SHADOW_GRID_HEAVY=$h RT_HIP_LIB=$PWD/build/variants/walk31.so python scripts/share_target.py 8 6 | tail -1; done
done
