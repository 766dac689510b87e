"""Diagnostic (GPU box): the UNPINNED gradient comparison of the headline image -- N = 2000, 80 classes, 16 blocks, seed 0, the image
tests/test_gpu_backward.py::test_headline_config_single_image accepts on the pinned piece -- and of the num_pwfeat_fc = 0 variant:
per parameter tensor max |g_hip - g_oracle| / max |g_oracle| with the oracle differentiating ITS OWN ReLU masks and winner sets.
Where a mask entry of the two forward passes differs (a pre-activation within fp32 noise of a kink) single gradient entries differ
by O(1): the numbers printed here say how much of the unpinned comparison that affects.  python tools/unpinned_stats.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import make_pair, make_image, grad_errors, gpu_pins  # noqa: E402
from oracle.pins import mask_disagreements  # noqa: E402

for nfc in (3, 0):
    n, c, b = 2000, 80, 16
    net, orc = make_pair(c, b, num_pwfeat_fc=nfc)
    net.keep_edge_activations = True
    batch = make_image(n, c, seed=0)
    ref, gref = orc.forward_backward(batch, keep=True)
    net.run(batch)
    torch.cuda.synchronize()
    n_diff, worst, where = mask_disagreements(gpu_pins(net), ref)
    un = grad_errors(net, gref, c, b, num_pwfeat_fc=nfc)
    _, gpin = orc.forward_backward(batch, pins=gpu_pins(net))
    pin = grad_errors(net, gpin, c, b, num_pwfeat_fc=nfc)
    v = np.array(list(un.values()))
    wk, wv = max(un.items(), key=lambda kv: kv[1])
    print("headline image (N=%d C=%d B=%d seed 0, num_pwfeat_fc=%d, E=%d): %d mask entries differ from the oracle's own (worst distance from the kink %.2e at %s)"
          % (n, c, b, nfc, int(net.num_edges), n_diff, worst, where))
    print("  UNPINNED: %d of %d tensors <= 2e-5, %d <= 1e-4, %d <= 1e-3; median %.2e; worst %.2e (%s)"
          % ((v <= 2e-5).sum(), v.size, (v <= 1e-4).sum(), (v <= 1e-3).sum(), float(np.median(v)), wv, wk))
    print("  pinned (the acceptance test): worst %.2e (%s)" % max((e, k) for k, e in pin.items()))
