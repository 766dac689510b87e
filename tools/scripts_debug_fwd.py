import sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, '..')
from tests.util import make_pair, rel_err, make_image
for (n, c, b, seed) in [(6, 1, 1, 0), (64, 80, 2, 2), (300, 80, 16, 0), (2000, 80, 16, 0)]:
    net, orc = make_pair(c, b)
    batch = make_image(n, c, seed=seed)
    ref = orc.forward(batch, with_loss=False, keep=True)
    net.run(batch, training=True, backward=False)
    torch.cuda.synchronize()
    pairs = net.neighbor_pair_idxs.cpu().numpy()
    print("case", n, c, b, "E", len(pairs), len(ref["neighbor_pair_idxs"]), "pairs_equal", np.array_equal(pairs, ref["neighbor_pair_idxs"]))
    if np.array_equal(pairs, ref["neighbor_pair_idxs"]):
        ious = ref["det_det_iou"][pairs[:, 0], pairs[:, 1]]
        print("  iou exact", np.array_equal(net.edge_iou.cpu().numpy(), ious))
        print("  pw", rel_err(net.pw_feats.cpu().numpy(), ref["pw_feats"].detach().numpy()))
        bf = net.block_feats
        print("  blocks", [float("%.2e" % rel_err(bf[k].cpu().numpy(), ref["block_feats"][k].detach().numpy())) for k in range(1, b + 1)])
        print("  pred", rel_err(net.prediction.cpu().numpy(), ref["prediction"].detach().numpy()), net.prediction[:4].cpu().numpy(), ref["prediction"][:4].detach().numpy())
    t0 = time.time()
    for _ in range(5):
        net.run(batch, training=True, backward=False)
    torch.cuda.synchronize()
    print("  fwd ms", (time.time() - t0) / 5 * 1e3)
