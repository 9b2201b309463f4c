#!/bin/bash
# The five profiled workloads of profiles/r03_*: `scripts/reprofile_r03.sh gpu` runs on the GPU box (through gpurun, ~12 minutes),
# `scripts/reprofile_r03.sh summarize` afterwards in the repository (rewrites profiles/r03_* and profiles/traffic.json, whose
# source hash has to match bench.kernel_source_sha16() for bench.py to report counter traffic).
set -e
case "$1" in
gpu)
  bash scripts/profile_r03.sh headline --steps 4 --warmup 1 &&
  bash scripts/profile_r03.sh bistro1080 --scene bistro_like --spp 16 --bounces 4 --steps 3 --warmup 1 &&
  bash scripts/profile_r03.sh cornell1080 --scene cornell12 --spp 64 --bounces 0 --steps 4 --warmup 1 &&
  bash scripts/profile_r03.sh bistro4kshare --scene bistro_like --width 3840 --height 2160 --spp 64 --bounces 8 --share-of 8 --steps 2 --warmup 1 &&
  bash scripts/profile_r03.sh reference --mode reference --steps 20 --warmup 3
  ;;
summarize)
  S=scripts/summarize_profile.py
  rm -f profiles/r03_extended_* profiles/r03_reference_* profiles/traffic.json
  python $S gpurun_out/prof_r03_headline profiles/r03_extended_sponza1080p_64spp_wavefront --frames 5 --workload sponza_like_1920x1080_extended_64spp_4b --match k_wf_ > /dev/null
  python $S gpurun_out/prof_r03_bistro1080 profiles/r03_extended_bistro1080p_16spp_wavefront --frames 4 --workload bistro_like_1920x1080_extended_16spp_4b --match k_wf_ > /dev/null
  python $S gpurun_out/prof_r03_cornell1080 profiles/r03_extended_cornell1080p_64spp_primary_single_pass --frames 5 --workload cornell12_1920x1080_extended_64spp_0b --match k_render_extended > /dev/null
  python $S gpurun_out/prof_r03_bistro4kshare profiles/r03_extended_bistro4k_64spp_8b_share8_wavefront --frames 3 --workload bistro_like_3840x2160_extended_64spp_8b_share8 --match k_wf_ > /dev/null
  python $S gpurun_out/prof_r03_reference profiles/r03_reference_sponza1080p_primary --frames 23 --workload sponza_like_1920x1080_reference --match k_render_reference > /dev/null
  for f in profiles/r03_*_pmc.json; do echo "== $f"; python scripts/issue_summary.py $f ${f%_pmc.json}_issue.json | grep "trace\|shade\|finish\|reference\|extended" || true; done
  python - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
t = json.load(open('profiles/traffic.json'))
print("sources:", bench.kernel_source_sha16())
for k, v in t.items():
    print(k, round(v['traffic'] / 1e9, 1), "GB", round(v['kernel_ms_per_frame'], 2), "ms", round(v['traffic'] / v['kernel_ms_per_frame'] / 1e6), "GB/s",
          v['dominant_kernel']['name'], round(v['dominant_kernel'].get('hbm_frac_of_8TBps') or 0, 3), v['source_sha16'])
PY
  ;;
*) echo "usage: $0 gpu|summarize"; exit 2;;
esac
