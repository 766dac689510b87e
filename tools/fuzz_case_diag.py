"""Diagnostic for one case of tools/fuzz_parity.py: is a gradient error at the fuzzer's bar the device's or fp32's?
Replays the fuzzer's random stream up to the case (no GPU work for the earlier ones), then compares the HIP gradient AND the fp32
oracle's pinned gradient with the fp64 twin of the oracle on the same smooth piece.   python tools/fuzz_case_diag.py <conf> <seed> <case>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "0", sys.argv[2], sys.argv[1], sys.argv[3]]
CASE = int(sys.argv[4])
import numpy as np, torch
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_parity.py")).read()
head = src[:src.index("t0 = time.time()")]
exec(compile(head, "fuzz_parity_head", "exec"))
for case in range(CASE + 1):
    pairs = [nasty_image() for _ in range(1 if rng.uniform() < 0.6 else int(rng.integers(2, 5)))]
imgs = [p[0] for p in pairs]
print("case", CASE, [(int(im["dets"].shape[0]), int(im["gt_boxes"].shape[0]), m) for im, m in pairs])
net.run(imgs if len(imgs) > 1 else imgs[0]); torch.cuda.synchronize()
o64 = go.GnetOracle(NC, NB, params={k: v.detach().numpy() for k, v in orc.params.items()}, dtype=torch.float64,
                    class_weights=orc.class_weights.numpy(), normalize_loss=orc.normalize_loss, pw_feat_multiplyer=orc.pw_feat_multiplyer,
                    neighbor_feats=NF, num_pwfeat_fc=NFC)
g32 = g64 = None
for i, im in enumerate(imgs):
    image = i if len(imgs) > 1 else None
    pins = gpu_pins(net, image)
    _, a = orc.forward_backward(im, pins=pins)
    _, b = o64.forward_backward(im, pins=pins)
    g32 = {k: np.asarray(v, np.float64) for k, v in a.items()} if g32 is None else {k: g32[k] + a[k] for k in g32}
    g64 = {k: np.asarray(v, np.float64) for k, v in b.items()} if g64 is None else {k: g64[k] + b[k] for k in g64}
rows = []
for name, _ in go.param_spec(NC, NB, None, NF, NFC):
    g = net.gradients[name].detach().cpu().numpy().reshape(-1).astype(np.float64)
    r64 = g64[name].reshape(-1); r32 = g32[name].reshape(-1)
    den = np.abs(r64).max() + 5e-2
    rows.append((np.abs(g - r64).max() / den, np.abs(r32 - r64).max() / den, np.abs(g - r32).max() / den, name))
rows.sort(reverse=True)
print("per tensor, relative to max |g64| + 0.05:   HIP vs fp64   fp32-oracle vs fp64   HIP vs fp32-oracle")
for r in rows[:6]:
    print("  %-50s %.2e   %.2e   %.2e" % (r[3], r[0], r[1], r[2]))
