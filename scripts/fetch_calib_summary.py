"""Summarise a `rocprofv3 --pmc FETCH_SIZE --kernel-trace` run of build/fetch_calib (scripts/micro/fetch_calib.hip) into
profiles/fetch_calibration.json: FETCH_SIZE (KiB) per kernel, bytes it reports per access, and the factor that turns the reported bytes
into the bytes the shape must bring from HBM (64-byte sectors of 128-byte lines: see `expected`).  summarize_profile.py reads
`gather_factor` for kernels whose reads are scattered 16-byte quads.
  python scripts/fetch_calib_summary.py gpurun_out/r03/fetch_calib profiles/r03_fetch_calibration.json"""
import csv, glob, json, os, re, sys

src, out = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(os.path.join(src, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1]
N = 4096 * 256 * 64
rows = []
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "FETCH_SIZE":
        continue
    rows.append((r["Kernel_Name"], float(r["Counter_Value"])))
res = {"accesses_per_gather_kernel": N, "table_bytes": 4 << 30, "unit": "FETCH_SIZE is reported in KiB", "kernels": []}
for name, kib in rows:
    m = re.search(r"k_gather<(\d+), (\d+)>", name)
    if "k_stream" in name:
        res["kernels"].append({"kernel": "k_stream", "fetch_size_KiB": kib, "bytes_read": 4 << 30, "reported_over_read": kib * 1024 / (4 << 30)})
    elif m:
        quads, align = int(m.group(1)), int(m.group(2))
        # 64-byte sectors a run of `quads` 16-byte quads starting at a multiple of `align` quads touches, averaged over start positions
        starts = [s * align % 4 for s in range(4)]
        sectors = sum(((st + quads - 1) // 4) + 1 for st in starts) / 4.0
        lines = sum((((s * align % 8) + quads - 1) // 8) + 1 for s in range(8)) / 8.0
        per = kib * 1024 / N
        res["kernels"].append({"kernel": f"gather {quads} quads aligned {align} quads", "fetch_size_KiB": kib, "reported_bytes_per_access": per,
                               "useful_bytes": quads * 16, "expected_64B_sectors": sectors, "expected_128B_lines": lines,
                               "reported_over_sector_bytes": per / (sectors * 64), "reported_over_line_bytes": per / (lines * 128)})
g = [k for k in res["kernels"] if k["kernel"].startswith("gather")]
if g:
    res["gather_factor_if_sectors"] = 1.0 / (sum(k["reported_over_sector_bytes"] for k in g) / len(g))
    res["gather_factor_if_lines"] = 1.0 / (sum(k["reported_over_line_bytes"] for k in g) / len(g))
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
