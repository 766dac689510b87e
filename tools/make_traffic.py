"""profiles/<round>_traffic.json from the FETCH_SIZE, WRITE_SIZE and SQ passes of tools/collect_profiles.sh.
usage: make_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <sq counter_collection.csv> <out.json>"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.pmc_sum import summarise  # noqa: E402
import bench  # noqa: E402

fetch, write, sq = summarise(sys.argv[1]), summarise(sys.argv[2]), summarise(sys.argv[3])
N_SIMD = 256 * 4
kernels = {}
for k in fetch:
    f = fetch[k].get("FETCH_SIZE", 0.0); w = write.get(k, {}).get("WRITE_SIZE", 0.0)
    kernels[k] = {"avg_launch_us": round(fetch[k]["avg_us"], 2), "launches": fetch[k]["launches"], "fetch_size_kb_raw": f,
                  "write_size_kb": w, "hbm_bytes": (2.0 * f + w) * 1024.0}
    q = sq.get(k)
    if q:
        busy, mf, va = q.get("VALU_MFMA_BUSY_CYCLES", 0.0), q.get("INSTS_MFMA", 0.0), q.get("INSTS_VALU", 0.0)
        # GPU-active cycles of the launch = its duration at the clock it really ran at (rocprofv3 reports the SUM over the
        # 8 XCDs' GRBM instances: 3.5e6 for a 192 us launch = 8 x 2.28 GHz x 192 us)
        gui = q.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        kernels[k].update({"sq_pass_avg_launch_us": round(q["avg_us"], 2), "valu_mfma_busy_cycles": busy, "insts_mfma": mf,
                           "insts_valu": va, "wait_inst_any": q.get("WAIT_INST_ANY"), "wait_inst_lds": q.get("WAIT_INST_LDS"),
                           "active_inst_any": q.get("ACTIVE_INST_ANY"), "wave_cycles": q.get("WAVE_CYCLES"), "gui_active_cycles": gui,
                           "mfma_busy": round(busy / (N_SIMD * gui), 4) if (busy and gui) else (0.0 if gui else None),
                           "effective_clock_ghz": round(gui / (q["avg_us"] * 1e3), 3) if gui else None,
                           # SQ_INSTS_VALU counts the MFMAs too (pw_fwd: 541 per wave and tile against 296 MFMAs + ~245 other
                           # vector instructions in the ISA): the ratio is of the OTHER vector instructions
                           "valu_per_mfma": round((va - mf) / mf, 2) if mf else None})
out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_* GRBM_GUI_ACTIVE (three separate passes) -- python bench.py --steps 2 "
                  "--warmup 1 --cpu-seconds 0 --no-kernel-timing --no-other-configs",
       "workload": [2000, 8, 80, 16, "dense"],
       "kernel_source_hash": bench.kernel_source_hash(),
       "units": "per launch; hbm_bytes = (2 x FETCH_SIZE[KB] + WRITE_SIZE[KB]) x 1024 -- MI355X_MICROARCH.md (HBM): on gfx950 "
                "FETCH_SIZE reports half of the bytes of wide coalesced reads (double it); other access widths and WRITE_SIZE "
                "are uncalibrated; Infinity-Cache hits are counted, not excluded.  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES (64 per "
                "v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_32x32x16_bf16, summed over the SIMDs) / (1024 SIMDs x GRBM_GUI_ACTIVE of the launch); valu_per_mfma = "
                "(SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA (INSTS_VALU includes the MFMAs: rounds 1-3 printed the ratio one too high)",
       "kernels": kernels}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps({k: [round(v["hbm_bytes"] / 1e6, 1), v.get("mfma_busy")] for k, v in kernels.items()}))
