"""Issue-side reading of a committed PMC summary (profiles/*_pmc.json from summarize_profile.py): per kernel, how busy the
VALU issue slots are, what the wave cycles are spent on, lane utilisation and occupancy.  SQ_* counters count quad-cycles
summed over the chip's 1024 SIMDs (MI355X_MICROARCH.md); clock taken as 2.4 GHz.
  python scripts/issue_summary.py profiles/X_pmc.json [out.json]"""
import json, sys
d = json.load(open(sys.argv[1]))
out = {"source": sys.argv[1], "clock_GHz_assumed": 2.4, "simds": 1024, "kernels": {}}
for k, c in d["pmc_per_frame_by_kernel"].items():
    ms = d["kernels"].get(k, {}).get("total_ms_per_frame")
    if not ms or "SQ_WAVE_CYCLES" not in c or ms < 0.1:
        continue
    simd_cycles = ms * 1e-3 * 2.4e9 * 1024
    wc, act = c["SQ_WAVE_CYCLES"], c.get("SQ_ACTIVE_INST_ANY", 0) or 1
    out["kernels"][k] = {
        "ms_per_frame": ms, "share_of_frame": ms / d["kernel_ms_per_frame"],
        "valu_issue_busy": 4 * c["SQ_ACTIVE_INST_VALU"] / simd_cycles,
        "valu_lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64) if "SQ_THREAD_CYCLES_VALU" in c else None,
        "waves_per_simd": 4 * wc / simd_cycles,
        "wave_cycles": {"waiting_SQ_WAIT_ANY": c["SQ_WAIT_ANY"] / wc, "issue_stalled_SQ_WAIT_INST_ANY": c.get("SQ_WAIT_INST_ANY", 0) / wc,
                        "issuing_SQ_ACTIVE_INST_ANY": c.get("SQ_ACTIVE_INST_ANY", 0) / wc},
        "issuing_split": {n: c.get("SQ_ACTIVE_INST_" + n, 0) / act for n in ("VALU", "SCA", "LDS", "FLAT", "MISC")},
        "instructions": {n: c.get("SQ_INSTS_" + n) for n in ("VALU", "SALU", "VMEM_RD", "VMEM_WR", "LDS", "BRANCH", "SMEM")},
        "hbm_GBps": c.get("hbm_GBps"),
    }
text = json.dumps(out, indent=1)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
for k, v in out["kernels"].items():
    print(f"{k:22s} {v['ms_per_frame']:7.2f} ms  VALU issue busy {v['valu_issue_busy']:.2f}  lanes {v['valu_lane_utilisation'] or 0:.2f}  waves/SIMD {v['waves_per_simd']:.1f}  "
          f"wait {v['wave_cycles']['waiting_SQ_WAIT_ANY']:.2f} stall {v['wave_cycles']['issue_stalled_SQ_WAIT_INST_ANY']:.2f} issue {v['wave_cycles']['issuing_SQ_ACTIVE_INST_ANY']:.2f}")
