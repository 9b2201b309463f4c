"""Turn a gpurun_out/prof_<tag>/ directory (scripts/profile_r02.sh) into the committed summaries under profiles/.

  python scripts/summarize_profile.py gpurun_out/prof_X profiles/NAME --frames N [--workload KEY] [--match k_wf_]

Writes <out>_kernel_stats.csv (verbatim rocprofv3 --kernel-trace --stats summary), <out>_pmc.json (per-FRAME counter
sums over every dispatch of the measured kernels: one frame of the wavefront pipeline is ~26 launches per batch) and
merges the HBM traffic per frame into profiles/traffic.json under KEY.
HBM bytes follow MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE are in KiB, collected in separate --pmc
passes; on gfx950 FETCH_SIZE reads half the bytes of a wide coalesced streaming read (x2).  For scattered 16-byte quads it does not:
profiles/r03_fetch_calibration.json (scripts/micro/fetch_calib.hip, round 3) - one request is counted as 64 bytes whether it brings a
32-byte sector, half a line or a whole 128-byte line.  So the read side is per kernel: x2 for the kernels that stream path state in
slot order (generate, shade, finish, resolve, the camera-list kernel), x1 for the kernels whose reads are gathers (k_wf_shadow_grid,
k_wf_trace): their FETCH_SIZE is a REQUEST count in units of 64 bytes, reported as such (`requests`)."""
import csv, glob, json, os, shutil, sys, collections

src, out = sys.argv[1], sys.argv[2]
arg = lambda k, d=None: sys.argv[sys.argv.index(k) + 1] if k in sys.argv else d
key, frames, match = arg("--workload"), int(arg("--frames", "1")), arg("--match", "k_render")
os.makedirs(os.path.dirname(out), exist_ok=True)
newest = lambda pattern: sorted(glob.glob(pattern), key=os.path.getmtime)  # gpurun merges runs into gpurun_out/: take the latest
stats = newest(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[-1]
shutil.copy(stats, out + "_kernel_stats.csv")
import re
rows = [r for r in csv.DictReader(open(stats)) if match in r["Name"] and "<true" not in r["Name"]]


def kname(full):
    n = re.search(r"(k_[a-z_0-9]+)", full).group(1)
    if n != "k_wf_trace":
        return n
    return n + ("<shadow>" if "<false, true>" in full else "<closest>" if "<false, false>" in full else "")


GATHER_KERNELS = ("k_wf_shadow_grid", "k_wf_trace<closest>", "k_wf_trace<shadow>", "k_wf_beams", "k_render_reference")
read_factor = lambda kn: 1.0 if kn in GATHER_KERNELS else 2.0


# bench.py renders warmup + steps frames plus ONE more with the counting kernel variants (<true, ...>, excluded
# here); kernels that are not templated on COUNT also run in that extra frame
nframes = lambda full: frames if "<" in full else frames + 1
summary = {"frames": frames, "kernels": {kname(r["Name"]): {"calls": int(r["Calls"]), "total_ms_per_frame": float(r["TotalDurationNs"]) / 1e6 / nframes(r["Name"]),
                                                             "avg_us": float(r["AverageNs"]) / 1e3} for r in rows}}
summary["kernel_ms_per_frame"] = sum(v["total_ms_per_frame"] for v in summary["kernels"].values())
pmc = {}
by_kernel = collections.defaultdict(dict)
for d in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_sq", "pmc_sq2", "pmc_sq3"):
    fs = newest(os.path.join(src, d, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(fs[-1])):
        if match in r["Kernel_Name"] and "<true" not in r["Kernel_Name"]:
            v = float(r["Counter_Value"]) / nframes(r["Kernel_Name"])
            agg[r["Counter_Name"]] += v
            kn = kname(r["Kernel_Name"])
            by_kernel[kn][r["Counter_Name"]] = by_kernel[kn].get(r["Counter_Name"], 0.0) + v
    for k, v in agg.items():
        pmc[k] = v
summary["pmc_per_frame"] = pmc
for kn, c in by_kernel.items(): # per-kernel HBM bytes (same unit / gfx950 correction as the totals below)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        c["read_factor"] = read_factor(kn)
        c["hbm_bytes"] = read_factor(kn) * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024
        c["read_requests_64B"] = c["FETCH_SIZE"] * 1024 / 64
        ms = summary["kernels"].get(kn, {}).get("total_ms_per_frame")
        if ms:
            c["hbm_GBps"] = c["hbm_bytes"] / ms / 1e6
            c["read_requests_per_s"] = c["read_requests_64B"] / ms * 1e3
summary["pmc_per_frame_by_kernel"] = by_kernel
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    rd, wr = pmc["FETCH_SIZE"] * 1024, pmc["WRITE_SIZE"] * 1024
    calibrated = sum(c.get("hbm_bytes", 0.0) for c in by_kernel.values())
    summary["hbm_bytes_per_frame"] = {"read_x1": rd, "read_x2_streaming_correction_everywhere": 2 * rd, "write": wr, "traffic": calibrated,
                                      "traffic_note": "per-kernel read factors: x2 streaming kernels, x1 gather kernels (profiles/r03_fetch_calibration.json)"}
    if key:
        # the entry bench.py reads: tied to the kernel sources it was measured on (bench.kernel_source_sha16)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        entry = {"traffic": calibrated, "traffic_if_x2_everywhere": 2 * rd + wr, "source_sha16": bench.kernel_source_sha16(), "profile": os.path.basename(out) + "_pmc.json",
                 "kernel_ms_per_frame": summary["kernel_ms_per_frame"]}
        dom = max(summary["kernels"].items(), key=lambda kv: kv[1]["total_ms_per_frame"])
        dk = by_kernel.get(dom[0], {})
        entry["dominant_kernel"] = {"name": dom[0], "ms_per_frame": dom[1]["total_ms_per_frame"], "share_of_frame": dom[1]["total_ms_per_frame"] / summary["kernel_ms_per_frame"],
                                    "launches_per_frame": dom[1]["calls"] / nframes(dom[0] + "<"), "avg_launch_ms": dom[1]["avg_us"] / 1e3,
                                    "hbm_bytes_per_frame": dk.get("hbm_bytes"), "hbm_bytes_per_launch": (dk.get("hbm_bytes") or 0) / max(1.0, dom[1]["calls"] / nframes(dom[0] + "<")) or None,
                                    "read_factor": dk.get("read_factor"), "read_requests_per_s": dk.get("read_requests_per_s"), "hbm_GBps": dk.get("hbm_GBps"),
                                    "hbm_frac_of_8TBps": (dk.get("hbm_GBps") or 0) / 8000.0 or None}
        if "SQ_ACTIVE_INST_VALU" in pmc:  # quad-cycles summed over the chip's 1024 SIMDs; 2.4 GHz
            entry["valu_busy"] = 4 * pmc["SQ_ACTIVE_INST_VALU"] / (summary["kernel_ms_per_frame"] * 1e-3 * 2.4e9 * 1024)
            if "SQ_ACTIVE_INST_VALU" in dk:
                entry["dominant_kernel"]["valu_busy"] = 4 * dk["SQ_ACTIVE_INST_VALU"] / (dom[1]["total_ms_per_frame"] * 1e-3 * 2.4e9 * 1024)
        tp = os.path.join(os.path.dirname(out), "traffic.json")
        t = json.load(open(tp)) if os.path.exists(tp) else {}
        t[key] = entry
        json.dump(t, open(tp, "w"), indent=1, sort_keys=True)
if "TCC_HIT_sum" in pmc:
    summary["l2_hit_rate"] = pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"])
for f in glob.glob(os.path.join(src, "bench_*.json")):
    try:
        line = [l for l in open(f).read().splitlines() if l.startswith("{")][-1]
        summary.setdefault("bench_lines", {})[os.path.basename(f)] = json.loads(line)
    except Exception:
        pass
json.dump(summary, open(out + "_pmc.json", "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "bench_lines"}, indent=1)[:3500])
