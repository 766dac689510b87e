import sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, '..')
from tests.util import make_pair, rel_err, make_image
from oracle import gnet_oracle as go
for (n, c, b, seed) in [(6, 1, 1, 0), (64, 80, 2, 2), (300, 80, 16, 0), (1000, 1, 16, 0)]:
    cw = np.linspace(0.5, 1.5, c + 1).astype(np.float32)
    net, orc = make_pair(c, b, class_weights=cw)
    batch = make_image(n, c, seed=seed)
    ref, gref = orc.forward_backward(batch)
    net.run(batch)
    torch.cuda.synchronize()
    print("case", n, c, b, "E", net.num_edges)
    print("  pred", rel_err(net.prediction.cpu().numpy(), ref["prediction"].detach().numpy()))
    print("  anno_iou exact", np.array_equal(net.det_anno_iou.cpu().numpy(), ref["det_anno_iou"]))
    print("  labels", np.array_equal(net.labels.cpu().numpy(), ref["labels"]), "assign", np.array_equal(net.det_gt_matching.cpu().numpy(), ref["det_gt_matching"]),
          "weights", rel_err(net.weights.cpu().numpy(), ref["weights"].numpy()), "npos", ref["labels"].sum())
    print("  loss", float(net.loss), float(ref["loss"]), "normed", float(net.loss_normed), float(ref["loss_normed"]))
    g = net.grads.cpu().numpy()
    off = 0
    worst = []
    for name, shape in go.param_spec(c, b):
        k = int(np.prod(shape))
        gr = gref[name].reshape(-1)
        e = rel_err(g[off:off + k], gr) if np.abs(gr).max() >= 1 else float(np.abs(g[off:off+k]-gr).max() / max(np.abs(gr).max(), 1e-30))
        worst.append((e, name, float(np.abs(gr).max())))
        off += k
    worst.sort(reverse=True)
    for w in worst[:6]:
        print("   grad", "%.2e" % w[0], w[1], "max|ref| %.3e" % w[2])
    print("  nan in grads", np.isnan(g).sum())
    t0 = time.time()
    for _ in range(5):
        net.run(batch)
    torch.cuda.synchronize()
    print("  fwd+bwd ms", (time.time() - t0) / 5 * 1e3)
