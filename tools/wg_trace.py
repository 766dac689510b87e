"""Per-workgroup time lines of the small kernels of a step (needs a GNET_TRACE build of the library):

    GNET_TRACE=1 python -m gossipnet_amd.build --force && GNET_TRACE=1 python tools/wg_trace.py [images] ; python -m gossipnet_amd.build --force

Every traced kernel writes wall_clock64() (100 MHz) stamps per workgroup: slot 0 = entry, 15 = exit, the others at its
phase boundaries (see the GSTAMP calls in csrc/).  Only the launches of block num_blocks / 2 are stamped.
Reported per kernel (microseconds): spread of the workgroups' entry times, entry -> each stamp (median / max over the
workgroups that have it), and first entry -> last exit (the part of the launch's duration that is inside the kernel)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KERNELS = ["PW_FWD", "EDGE_FWD", "NODE_FWD", "EDGE_BWD", "GATHER", "NODE_BWD", "PW_BWD"]
MAXWG = 8192
bufs = {}
for k in KERNELS:
    bufs[k] = torch.zeros(MAXWG * 16, dtype=torch.int64, device="cuda")
    os.environ["GNET_TRACE_" + k] = str(bufs[k].data_ptr())
from gossipnet_amd.config import cfg, experiment_cfg  # noqa: E402
from gossipnet_amd.network import Gnet, DeviceBatch  # noqa: E402
from gossipnet_amd.synthetic import make_image  # noqa: E402

images = int(sys.argv[1]) if len(sys.argv) > 1 else 1
experiment_cfg()
net = Gnet(80, device=torch.device("cuda"))
b = DeviceBatch([make_image(2000, 80, seed=i, preset="dense") for i in range(images)], torch.device("cuda"))
for _ in range(4):
    net.run(b)
torch.cuda.synchronize()
print("images %d  N %d  E %d" % (images, net.num_dets, net.num_edges))
for k in KERNELS:
    d = bufs[k].cpu().numpy().reshape(MAXWG, 16).astype(np.float64) / 100.0     # -> microseconds
    live = d[:, 0] > 0
    d = d[live]
    if not len(d):
        print(k, "no stamps"); continue
    t0 = d[:, 0].min()
    ent = d[:, 0] - t0
    print("%s: %d workgroups; entry spread p50 %.1f p90 %.1f max %.1f us; first entry -> last exit %.1f us"
          % (k, len(d), np.percentile(ent, 50), np.percentile(ent, 90), ent.max(), d[:, 15].max() - t0))
    for s in range(1, 16):
        have = d[:, s] > 0
        if have.any():
            rel = d[have, s] - d[have, 0]
            print("   slot %2d: %4d wgs, entry -> stamp median %6.2f  p90 %6.2f  max %6.2f us;  first entry -> stamp max %6.2f"
                  % (s, have.sum(), np.median(rel), np.percentile(rel, 90), rel.max(), d[have, s].max() - t0))
    # per-XCD picture (workgroups are dealt round-robin to the 8 XCDs): exit time relative to the first entry
    if len(d) >= 64:
        idx = np.nonzero(live)[0]
        ex = d[:, 15] - t0
        print("   exit by XCD (blockIdx % 8): " + "  ".join("%d: %.0f/%.0f" % (x, np.median(ex[idx % 8 == x]), ex[idx % 8 == x].max()) for x in range(8)) + "  (median/max us)")
        G = int(idx.max()) + 1
        if G % 8 == 0:
            lb = (idx % 8) * (G // 8) + idx // 8          # the kernels' XCD-aware logical index = position in the edge list
            order = np.argsort(lb)
            chunks = np.array_split(ex[order], 16)
            print("   exit along the edge list (16 slices of the logical index, median): " + " ".join("%.0f" % np.median(c) for c in chunks))
        # same CU? workgroups b and b + 8*k share an XCD; spread within slices tells hardware from data effects
        if os.environ.get("WG_TRACE_DETAIL") == k:
            s1 = d[:, 1] - d[:, 0]
            order = np.argsort(idx)
            print("   slot 1 (front) by dispatch order, 24 slices, median/max: " + " ".join("%.1f/%.1f" % (np.median(c), c.max()) for c in np.array_split(s1[order], 24)))
            o = idx // 8                                    # dispatch order within the XCD
            oo = np.argsort(o, kind="stable")
            print("   exit by dispatch order within the XCD (blockIdx // 8), 12 slices, median/max: " + " ".join("%.0f/%.0f" % (np.median(c), c.max()) for c in np.array_split(ex[oo], 12)))
            for sl in (5, 9, 13, 14):
                if (d[:, sl] > 0).all():
                    print("   slot %d by dispatch order, 12 slices, median: " % sl + " ".join("%.0f" % np.median(c) for c in np.array_split((d[:, sl] - t0)[oo], 12)))
            late = idx[s1 > np.percentile(s1, 85)]
            print("   late workgroups: blockIdx %% 8 histogram %s; blockIdx // 8 quartiles %s" % (np.bincount(late % 8, minlength=8).tolist(), np.percentile(late // 8, [0, 25, 50, 75, 100]).tolist()))
