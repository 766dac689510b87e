import sys
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, '..')
from tests.util import make_pair, rel_err, make_image
from oracle import gnet_oracle as go
c, b, n = 80, 1, 64
net, orc = make_pair(c, b)
for seed in range(4):
    batch = make_image(n, c, seed=seed)
    st = {}
    ref, gref = orc.forward_backward(batch, stats=st)
    net.run(batch); torch.cuda.synchronize()
    x1 = net.block_feats[1].cpu().numpy(); xr = ref["block_feats"][1].detach().numpy()
    q = net.debug_view("blk_q", n * 64, index=1).cpu().numpy().reshape(n, 64)
    pm = net.debug_view("blk_pm", n * 64, dtype=torch.int64, index=1).cpu().numpy().reshape(n, 64)
    pmax = (pm >> 32).astype(np.uint32).view(np.float32); cnt = pm & 0xffffffff
    print("seed", seed, "mask mismatch x1:", int(((x1 > 0) != (xr > 0)).sum()), "min|x1| nz", float(np.abs(xr[xr != 0]).min()),
          "cnt>1 where p>0:", int(((cnt > 1) & (pmax > 0)).sum()), "relu_margin", st["relu_margin"], "gap", st["max_gap"])
    # check ties in oracle
    g = net.grads.cpu().numpy()
    off = 0
    for name, shape in go.param_spec(c, b):
        k = int(np.prod(shape)); gr = gref[name].reshape(-1)
        e = float(np.abs(g[off:off+k]-gr).max() / max(np.abs(gr).max(), 1e-30)); off += k
        if "block1/fc2/biases" in name or "fc1/biases" in name and "block1" in name or "pw_fc2/biases" in name:
            print("    ", name, "%.2e" % e)
