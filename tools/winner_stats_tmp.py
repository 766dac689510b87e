import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gossipnet_amd.config import cfg, reset_cfg
from gossipnet_amd.network import Gnet, DeviceBatch
from gossipnet_amd.synthetic import make_image
reset_cfg()
dev = torch.device("cuda")
for preset in ("dense", "coco_like"):
    net = Gnet(80, device=dev)
    imgs = [make_image(2000, 80, seed=i, preset=preset) for i in range(2)]
    b = DeviceBatch(imgs, dev)
    net.run(b); torch.cuda.synchronize()
    E = net.num_edges; N = 4000
    ec = net._view(net._buf.edge_c, E, torch.int32).long()
    for blk in (1, 8, 16):
        h1 = net._view(net._buf.blk_h1[blk], E * 64, torch.float32).view(E, 64)
        W2 = net.variables["gnet/block%d/pw_fc2/weights" % blk]; b2 = net.variables["gnet/block%d/pw_fc2/biases" % blk]
        h2 = torch.relu(h1 @ W2 + b2)
        # segment argmax per (centre, col)
        mx = torch.zeros(N, 64, device=dev).scatter_reduce(0, ec[:, None].expand(E, 64), h2, "amax", include_self=True)
        is_max = (h2 == mx[ec]) & (h2 > 0)
        win = is_max.any(1)
        wc = torch.zeros(N, device=dev).scatter_add(0, ec, win.float())
        deg = torch.bincount(ec, minlength=N).float()
        print(preset, "block", blk, "E/N %.1f" % (E / N), "winner rows %.3f of E" % (win.float().mean().item()), "W_c mean %.1f max %d" % (wc.mean().item(), int(wc.max().item())),
              "nnz per winner row %.2f" % (is_max.sum().item() / max(1, win.sum().item())), "frac (c,j) with max>0 %.2f" % ((mx > 0).float().mean().item()))
