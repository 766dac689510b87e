"""python tools/roi_bench.py -- the roi_pool entry of bench.py on its own."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
print(json.dumps(bench.roi_pool_bench(torch.device("cuda", 0)), indent=1))
