import sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, '..')
from gossipnet_amd.config import cfg, reset_cfg
from gossipnet_amd.network import Gnet, DeviceBatch
from gossipnet_amd.synthetic import make_image
from gossipnet_amd import _lib
reset_cfg(); net = Gnet(80)
batch = DeviceBatch([make_image(2000, 80, seed=i) for i in range(8)], net.device)
for _ in range(2): net.run(batch)
torch.cuda.synchronize()
lib = _lib.load(); buf = (C.c_longlong * 16)()
lib.gnet_debug_prof(buf, 1)
net.run(batch); torch.cuda.synchronize()
lib.gnet_debug_prof(buf, 0)
v = np.array(list(buf), dtype=np.float64)
names = ["looptop", "h1 init", "L1 mfma", "P prefetch", "lds write+sync", "rn prefetch", "L2 mfma", "segment"]
tot = v[:8].sum()
for n_, x in zip(names, v): print('%-14s %10.0f  %5.1f%%' % (n_, x / 16, 100 * x / tot))
print('total cycles per launch (wave)', tot / 16, 'E', net.num_edges)
