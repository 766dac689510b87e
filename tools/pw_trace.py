"""Phase time stamps of pw_bwd_bf's sixth tile per workgroup (needs a GNET_TRACE build: GNET_TRACE=1 python -m gossipnet_amd.build &&
GNET_TRACE=1 python tools/pw_trace.py).  Slots: 1 tile top, 2 after d2 / scatter / dW3, 3 after the dW2 loop, 4 after the vmcnt(0)
in front of the ring, 5 behind barrier A, 6 after the d h1 loop, 7 behind barrier B; 8-14 the same from wave 4 (wave 0's partner
on its SIMD).  Prints the median / p90 over the workgroups of every interval, microseconds."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
buf = torch.zeros(8192 * 16, dtype=torch.int64, device="cuda")
os.environ["GNET_TRACE_PW_BWD"] = str(buf.data_ptr())
buf_f = torch.zeros(8192 * 16, dtype=torch.int64, device="cuda")
os.environ["GNET_TRACE_PW_FWD"] = str(buf_f.data_ptr())
from gossipnet_amd.config import cfg, experiment_cfg  # noqa: E402
from gossipnet_amd.network import Gnet, DeviceBatch  # noqa: E402
from gossipnet_amd.synthetic import make_image  # noqa: E402
experiment_cfg()
net = Gnet(80, device=torch.device("cuda"))
b = DeviceBatch([make_image(2000, 80, seed=i, preset="dense") for i in range(8)], torch.device("cuda"))
for _ in range(4):
    net.run(b)
torch.cuda.synchronize()
d = buf.cpu().numpy().reshape(8192, 16).astype(np.float64) / 100.0
d = d[d[:, 0] > 0]
print("workgroups", len(d), " kernel (first entry -> last exit) %.1f us" % (d[:, 15].max() - d[:, 0].min()))
names = {1: "tile top", 2: "prep done (d2, scatter, dW3)", 3: "dW2 loop done", 4: "vmcnt(0) done", 5: "behind barrier A", 6: "d h1 loop done", 7: "behind barrier B"}
for base, who in ((1, "wave 0"), (8, "wave 4")):
    print(who)
    for k in range(1, 7):
        dt = d[:, base + k] - d[:, base + k - 1]
        print("   %-32s -> %-32s median %6.2f  p90 %6.2f us" % (names[k], names[k + 1], np.median(dt), np.percentile(dt, 90)))
    tot = d[:, base + 6] - d[:, base]
    print("   tile: median %.2f p90 %.2f us" % (np.median(tot), np.percentile(tot, 90)))
print("wave 4 - wave 0 at each stamp (median): " + " ".join("%.2f" % np.median(d[:, 8 + k] - d[:, 1 + k]) for k in range(7)))

# pw_fwd3: 1 tile top, 2 before fc1 of the next tile (k-step 5), 3 after it, 4 k-step 12, 5 fc2 done, 6 epilogue done (before the barrier), 7 behind it
d = buf_f.cpu().numpy().reshape(8192, 16).astype(np.float64) / 100.0
d = d[d[:, 0] > 0]
print("pw_fwd3: workgroups", len(d), " kernel %.1f us" % (d[:, 15].max() - d[:, 0].min() if (d[:, 15] > 0).any() else -1))
nf = {1: "tile top", 2: "k-step 5 (before fc1)", 3: "fc1 of the next tile done", 4: "k-step 12", 5: "fc2 done", 6: "epilogue done", 7: "behind the barrier"}
for base, who in ((1, "wave 0"), (8, "wave 4")):
    print(who)
    for k in range(1, 7):
        dt = d[:, base + k] - d[:, base + k - 1]
        print("   %-28s -> %-28s median %6.2f  p90 %6.2f us" % (nf[k], nf[k + 1], np.median(dt), np.percentile(dt, 90)))
    tot = d[:, base + 6] - d[:, base]
    print("   tile: median %.2f p90 %.2f us" % (np.median(tot), np.percentile(tot, 90)))
if os.environ.get("PBB_TRACE2"):
    d = buf.cpu().numpy().reshape(8192, 16).astype(np.float64) / 100.0
    d = d[d[:, 0] > 0]
    seq = [(1, "tile top"), (8, "DMA issued"), (9, "d3 sources requested"), (10, "ids staged / loaded"), (11, "d2 MFMAs issued"), (12, "mask + sums done"), (2, "split done (prep done)")]
    print("phase A front, wave 0 (PBB_TRACE2):")
    for (sa, na), (sb, nb) in zip(seq[:-1], seq[1:]):
        dt = d[:, sb] - d[:, sa]
        print("   %-26s -> %-26s median %6.2f p90 %6.2f us" % (na, nb, np.median(dt), np.percentile(dt, 90)))
