import sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, '..')
from tests.util import make_pair, rel_err, make_image
from oracle import gnet_oracle as go
for (n, c, b, seed) in [(20, 1, 1, 0), (64, 1, 1, 2), (64, 1, 2, 2), (64, 80, 1, 2), (64, 80, 2, 2), (200, 1, 1, 3), (33,1,2,4)]:
    net, orc = make_pair(c, b)
    batch = make_image(n, c, seed=seed)
    ref, gref = orc.forward_backward(batch)
    net.run(batch)
    torch.cuda.synchronize()
    g = net.grads.cpu().numpy()
    off = 0
    res = []
    for name, shape in go.param_spec(c, b):
        k = int(np.prod(shape))
        gr = gref[name].reshape(-1)
        e = float(np.abs(g[off:off+k]-gr).max() / max(np.abs(gr).max(), 1e-30))
        res.append((name.replace("gnet/",""), e))
        off += k
    print("case", n, c, b, "E", net.num_edges, "tiles", (net.num_edges+31)//32)
    print("   ", "  ".join("%s:%.1e" % (nm, e) for nm, e in res))
