"""profiles/<round>_traffic.json from the FETCH_SIZE and WRITE_SIZE passes of tools/collect_profiles.sh.
usage: make_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.pmc_sum import summarise  # noqa: E402
import bench  # noqa: E402

fetch, write = summarise(sys.argv[1]), summarise(sys.argv[2])
kernels = {}
for k in fetch:
    f = fetch[k].get("FETCH_SIZE", 0.0); w = write.get(k, {}).get("WRITE_SIZE", 0.0)
    kernels[k] = {"avg_launch_us": round(fetch[k]["avg_us"], 2), "launches": fetch[k]["launches"], "fetch_size_kb_raw": f,
                  "write_size_kb": w, "hbm_bytes": (2.0 * f + w) * 1024.0}
out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --steps 2 --warmup 1 --cpu-seconds 0 "
                  "--no-kernel-timing --no-other-configs",
       "workload": [2000, 8, 80, 16, "dense"],
       "kernel_source_hash": bench.kernel_source_hash(),
       "units": "bytes per launch; hbm_bytes = (2 x FETCH_SIZE[KB] + WRITE_SIZE[KB]) x 1024 -- MI355X_MICROARCH.md (HBM): on gfx950 "
                "FETCH_SIZE reports half of the bytes of wide coalesced reads (double it); other access widths and WRITE_SIZE "
                "are uncalibrated; Infinity-Cache hits are counted, not excluded",
       "kernels": kernels}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes"] / 1e6, 1) for k, v in kernels.items()}))
