"""Turn a gpurun_out/prof_<tag>/ directory (scripts/profile_r01.sh) into the committed summaries under profiles/.

  python scripts/summarize_profile.py gpurun_out/prof_r01b profiles/r01_extended_sponza1080p_64spp [--workload KEY]

Writes <out>_kernel_stats.csv (verbatim rocprofv3 --kernel-trace --stats summary), <out>_pmc.json (per-launch
counter means for the dominant kernel) and merges the HBM traffic into profiles/traffic.json under KEY.
HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads half the
bytes of a wide coalesced streaming read, so the read side is reported as a [x1, x2] bracket with the x2 value
used for `traffic` (conservative: more traffic, lower efficiency)."""
import csv, glob, json, os, shutil, sys, collections

src, out = sys.argv[1], sys.argv[2]
key = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else None
os.makedirs(os.path.dirname(out), exist_ok=True)
stats = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, out + "_kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
dom = max((r for r in rows if "k_render" in r["Name"] and "<true>" not in r["Name"]), key=lambda r: float(r["TotalDurationNs"]))
summary = {"kernel": dom["Name"], "calls": int(dom["Calls"]), "avg_ns": float(dom["AverageNs"]), "min_ns": float(dom["MinNs"]), "max_ns": float(dom["MaxNs"])}
kname = dom["Name"]
pmc = {}
for d in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_sq"):
    fs = glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r["Kernel_Name"] == kname:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            pmc.setdefault("_meta", {"VGPR_Count": r["VGPR_Count"], "SGPR_Count": r["SGPR_Count"], "LDS_Block_Size": r["LDS_Block_Size"],
                                     "Scratch_Size": r["Scratch_Size"], "Grid_Size": r["Grid_Size"], "Workgroup_Size": r["Workgroup_Size"]})
    for k, v in agg.items():
        pmc[k] = {"mean": sum(v) / len(v), "min": min(v), "max": max(v), "n": len(v)}
summary["pmc"] = pmc
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    rd, wr = pmc["FETCH_SIZE"]["mean"] * 1024, pmc["WRITE_SIZE"]["mean"] * 1024
    summary["hbm_bytes_per_launch"] = {"read_x1": rd, "read_x2_gfx950_corrected": 2 * rd, "write": wr, "traffic": 2 * rd + wr}
    if key:
        tp = os.path.join(os.path.dirname(out), "traffic.json")
        t = json.load(open(tp)) if os.path.exists(tp) else {}
        t[key] = 2 * rd + wr
        json.dump(t, open(tp, "w"), indent=1, sort_keys=True)
if "TCC_HIT_sum" in pmc:
    h, m = pmc["TCC_HIT_sum"]["mean"], pmc["TCC_MISS_sum"]["mean"]
    summary["l2_hit_rate"] = h / (h + m)
for f in glob.glob(os.path.join(src, "bench_*.json")):
    try:
        line = [l for l in open(f).read().splitlines() if l.startswith("{")][-1]
        summary.setdefault("bench_lines", {})[os.path.basename(f)] = json.loads(line)
    except Exception:
        pass
json.dump(summary, open(out + "_pmc.json", "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "bench_lines"}, indent=1)[:3000])
