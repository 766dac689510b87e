"""Quick timing of the two step shapes (no kernel table, no CPU baseline): python tools/quick_bench.py [steps]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for images in (1, 8):
    r = bench.time_config(dev, 80, 16, 2000, images, "dense", steps if images == 1 else max(10, steps // 2), 5)
    print("images/step %d: %.1f det/s  %.4f ms/step  E/N %.1f" % (images, r["detections_per_sec"], r["ms_per_step"], r["edges_per_det"]))
