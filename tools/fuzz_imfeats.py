"""Fuzz of the image-feature variant (crop_windows -> RoI pooling -> reduce_imfeats FCs -> block_feats[0], and back): feature
maps of odd sizes, detections hanging over the map's edges, maps smaller than a crop, several images with different maps,
channel counts on both sides of the RoI kernels' 256-channel paths, imfeats_need_grad on and off -- no fault, finite results;
every case small enough is compared with the CPU oracle (roifeats bit-exact, outputs and pinned gradients <= 1e-5).
python tools/fuzz_imfeats.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gossipnet_amd.config import cfg, experiment_cfg
from gossipnet_amd.network import Gnet
from gossipnet_amd.synthetic import make_image
from oracle import gnet_oracle as go
from tests.util import rel_err, gpu_pins

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
NC, NB = 80, 2
t0 = time.time()
compared = 0
for case in range(cases):
    ch = int(rng.choice([4, 32, 64, 256, 512]))          # (gnet_fc_*: K = 49 ch and N multiples of 4, else GNET_ERR_UNSUPPORTED)
    dim = int(rng.choice([0, 64, 128]))
    imf = {"channels": ch, "imfeat_dim": dim, "crop": 7, "stride": 16}
    experiment_cfg()
    cfg.gnet.num_blocks = NB
    cfg.gnet.imfeats = True
    cfg.gnet.imfeat_dim = dim
    params = go.init_params(NC, NB, imfeat=imf, seed=int(rng.integers(1000)))
    net = Gnet(NC, imfeat_channels=ch, imfeat_stride=16)
    net.keep_edge_activations = True
    net.load_params(params)
    net.imfeats_need_grad = bool(rng.integers(2))
    n_img = int(rng.integers(1, 4))
    imgs = []
    for _ in range(n_img):
        im = make_image(int(rng.choice([1, 2, 31, 33, 64, 90, 130])), NC, seed=int(rng.integers(1 << 30)))
        H, W = int(rng.integers(1, 45)), int(rng.integers(1, 45))
        im["imfeats"] = rng.normal(size=(1, H, W, ch)).astype(np.float32)
        if rng.uniform() < 0.3:                       # detections far outside the map
            im["dets"] = im["dets"] + np.float32(rng.choice([-900.0, 900.0]))
        imgs.append(im)
    desc = [(int(im["dets"].shape[0]),) + im["imfeats"].shape[1:] for im in imgs]
    try:
        net.run(imgs if n_img > 1 else imgs[0])
        torch.cuda.synchronize()
        assert bool(torch.isfinite(net.grads).all().item()) and np.isfinite(net.loss.cpu().numpy()).all()
        if n_img == 1 and ch <= 64:
            orc = go.GnetOracle(NC, NB, params=params, imfeat=imf)
            ref = orc.forward(imgs[0])
            assert np.array_equal(net.roifeats.cpu().numpy(), ref["roifeats"]), "roifeats"
            assert rel_err(net.prediction.cpu().numpy(), ref["prediction"].detach().numpy()) < 1e-5
            _, gpin = orc.forward_backward(imgs[0], pins=gpu_pins(net))
            for name, _shape in go.param_spec(NC, NB, imf):
                g = net.gradients[name].detach().cpu().numpy().reshape(-1).astype(np.float64)
                gr = np.asarray(gpin[name], np.float64).reshape(-1)
                e = float(np.abs(g - gr).max() / (np.abs(gr).max() + 1e-2))
                assert e <= 1e-5, (name, e)
            compared += 1
    except Exception as e:
        print("case %d ch %d dim %d (dets, H, W, C) %s: %s: %s" % (case, ch, dim, desc, type(e).__name__, e), flush=True)
        raise
    if os.environ.get("FUZZ_VERBOSE"):
        print("case", case, ch, dim, desc, flush=True)
experiment_cfg()
print("imfeats fuzz: %d cases (%d compared with the oracle) in %.1f s, no fault" % (cases, compared, time.time() - t0))
