"""Training-step update on the GPU against the numpy restatement of TF's rules."""
import numpy as np
import pytest
import torch

from oracle import optim_oracle as oo
from tests.util import make_pair, make_image

pytestmark = pytest.mark.gpu


def test_adam_momentum_clip_against_oracle():
    from gossipnet_amd.config import cfg
    from gossipnet_amd.train import Optimizer
    net, _ = make_pair(1, 1)
    offs = np.concatenate([[0], np.cumsum([int(np.prod(s)) for _, s in net._spec])])
    rng = np.random.default_rng(0)
    for kind, clip in (("adam", -1.0), ("adam", 0.05), ("sgd", 0.5)):
        cfg.train.optimizer = kind
        cfg.train.gradient_clipping = clip
        opt = Optimizer(net)
        p = net.params.cpu().numpy().astype(np.float64)
        m = np.zeros_like(p); v = np.zeros_like(p)
        for t in range(1, 4):
            g = rng.normal(size=p.shape).astype(np.float32) * 0.01
            net.grads.copy_(torch.from_numpy(g).to(net.device))
            opt.apply_gradients(1e-3)
            gg = oo.clip_by_norm(g, offs, clip) if clip > 0 else g.astype(np.float64)
            if kind == "adam":
                p, m, v = oo.adam_step(p, gg, m, v, 1e-3, t)
            else:
                p, m = oo.momentum_step(p, gg, m, 1e-3, cfg.train.momentum)
            got = net.params.cpu().numpy()
            assert np.abs(got - p).max() < 2e-6, (kind, clip, t)
    cfg.train.optimizer = "adam"; cfg.train.gradient_clipping = -1.0


def test_train_steps_reduce_the_loss():
    from gossipnet_amd.config import cfg
    from gossipnet_amd.train import Optimizer, LearningRate, ExponentialMovingAverage, train_step
    net, _ = make_pair(80, 2)
    net.weight_reg = cfg.train.weight_decay
    cfg.train.lr_multi_step = [(3, 1e-3), (100, 1e-4)]
    opt, lr, ema = Optimizer(net), LearningRate(), ExponentialMovingAverage(0.7)
    batch = make_image(150, 80, seed=0)
    losses = []
    for it in range(1, 9):
        losses.append(float(train_step(net, opt, batch, lr.get_lr(it))))
        ema.apply(loss=losses[-1])
    assert losses[-1] < losses[0]
    assert lr.get_lr(50) == 1e-4 and opt.global_step == 8
    assert min(losses) <= ema.average("loss") <= max(losses)


def test_learning_rate_schedule_cpu_semantics():
    from gossipnet_amd.config import cfg
    from gossipnet_amd.train import LearningRate
    cfg.train.lr_multi_step = [(2, 0.1), (4, 0.01)]
    lr = LearningRate()
    assert [lr.get_lr(i) for i in range(1, 7)] == [0.1, 0.1, 0.01, 0.01, 0.01, 0.01]   # train.py:31-37


def test_one_rank_nccl_step_equals_single_process_step(tmp_path):
    """The product gradient buffer meets torch.distributed before the 8-GPU node does: a 1-rank `nccl` (= RCCL) group,
    broadcast_parameters(net.params), train_step(..., dist=dist) with the all-reduce on the flat net.grads buffer --
    bitwise the same parameters as the step without a process group."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = """
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
from gossipnet_amd.config import cfg, experiment_cfg
from gossipnet_amd.network import Gnet
from gossipnet_amd.synthetic import make_image
from gossipnet_amd.train import Optimizer, train_step
from gossipnet_amd.data_parallel import broadcast_parameters, shard_images
use_dist = sys.argv[2] == "1"
dist = None
if use_dist:
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
experiment_cfg(); cfg.gnet.num_blocks = 3
net = Gnet(80, weight_reg=0.0005)
if use_dist: broadcast_parameters(net.params, dist)
opt = Optimizer(net)
imgs = [make_image(n, 80, seed=s) for n, s in ((150, 0), (90, 1), (200, 2))]
costs = [float(Gnet.count_edges(im["dets"], "cuda:0")) for im in imgs]
mine = shard_images(imgs, 0, 1, costs=costs, per_rank=3)
net.grad_scale = 1.0 / len(imgs)
for it in range(3):
    train_step(net, opt, mine, 1e-3, dist=dist)
torch.cuda.synchronize()
np.save(sys.argv[1], net.params.cpu().numpy())
if use_dist: dist.destroy_process_group()
""" % root
    outs = []
    for mode in ("0", "1"):
        f = str(tmp_path / ("p%s.npy" % mode))
        subprocess.run([sys.executable, "-c", script, f, mode], check=True, cwd=root, timeout=600)
        outs.append(np.load(f))
    assert np.isfinite(outs[0]).all() and np.array_equal(outs[0], outs[1])


def test_two_ranks_rccl_two_gpus_equal_the_single_process_step(tmp_path):
    """The same comparison as the next test, over RCCL: two ranks on two GPUs under the `nccl` backend (one process per
    GPU, device = rank).  Needs two GPUs: on the one-GPU boxes of this pool it is SKIPPED and says so."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs for a 2-rank RCCL group; this box has %d" % torch.cuda.device_count())
    _two_ranks_equal_single_process(tmp_path, 5, "nccl")


@pytest.mark.parametrize("n_images", [5, 1])
def test_two_ranks_on_one_gpu_equal_the_single_process_step(tmp_path, n_images):
    _two_ranks_equal_single_process(tmp_path, n_images, "gloo")


def _two_ranks_equal_single_process(tmp_path, n_images, backend):
    """N > 1 on the PRODUCT path before an 8-GPU node ever runs it: two processes share cuda:0 under a `gloo` group
    (it accepts device tensors); each runs train_step(net, opt, shard, lr, dist=dist) on its LPT-by-edge-count shard of
    a 5-image global step -- HIP gradients into the flat buffer, ONE all-reduce, grad_scale = 1 / images of the step,
    reg_scale = 1 / world, parameters broadcast from rank 0 -- and after two optimizer steps both replicas hold the
    parameters of the single-process run over all five images (reduction order differs: <= 1e-6 relative).  With ONE image
    in the global step the second rank's shard is empty: it contributes zeros and still ends with rank 0's parameters."""
    import subprocess, sys, os, socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = """
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
from gossipnet_amd.config import cfg, experiment_cfg
from gossipnet_amd.network import Gnet
from gossipnet_amd.synthetic import make_image
from gossipnet_amd.train import Optimizer, train_step
from gossipnet_amd.data_parallel import broadcast_parameters, shard_images
out, rank, world, port, backend = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
dist = None
dev = rank if backend == "nccl" else 0
torch.cuda.set_device(dev)
if world > 1:
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
experiment_cfg(); cfg.gnet.num_blocks = 3
cfg.train.optimizer = "sgd"                     # momentum update: linear in the gradient (Adam's g / sqrt(v) turns a
                                                # 1e-7 reduction-order difference of a cancelling sum into a visible step)
cfg.random_seed = 42 + rank                     # replicas start DIFFERENT: the broadcast must make them equal
net = Gnet(80, weight_reg=0.0005, device="cuda:%%d" %% dev)
if world > 1: broadcast_parameters(net.params, dist, src=0)
opt = Optimizer(net)
imgs = [make_image(n, 80, seed=s) for n, s in ((150, 0), (90, 1), (200, 2), (60, 3), (120, 4))][:%d]
costs = [float(Gnet.count_edges(im["dets"], "cuda:%%d" %% dev)) for im in imgs]
mine = shard_images(imgs, rank, world, costs=costs)
assert 1 <= len(mine) < len(imgs) or world == 1 or len(imgs) == 1
net.grad_scale = 1.0 / len(imgs)
for it in range(2):
    train_step(net, opt, mine, 1e-2, dist=dist)
torch.cuda.synchronize()
np.save(out, net.params.cpu().numpy())
if world > 1:
    dist.barrier(); dist.destroy_process_group()
""" % (root, n_images)
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = str(s_.getsockname()[1]); s_.close()
    single = str(tmp_path / "single.npy")
    subprocess.run([sys.executable, "-c", script, single, "0", "1", port, backend], check=True, cwd=root, timeout=600)
    files = [str(tmp_path / ("r%d.npy" % r)) for r in range(2)]
    procs = [subprocess.Popen([sys.executable, "-c", script, files[r], str(r), "2", port, backend], cwd=root) for r in range(2)]
    for p_ in procs:
        assert p_.wait(timeout=600) == 0
    want = np.load(single)
    r0, r1 = np.load(files[0]), np.load(files[1])
    assert np.array_equal(r0, r1), "replicas hold identical parameters after the all-reduce"
    assert np.isfinite(want).all() and np.abs(r0 - want).max() <= 1e-6 * np.abs(want).max()
    from gossipnet_amd.config import experiment_cfg
    experiment_cfg()
    start = make_pair(80, 3)[0]            # (same seed-42 initialisation as rank 0: the steps did move the parameters)
    assert np.abs(want - start.params.cpu().numpy()).max() > 1e-4


def test_bench_two_ranks_functional(tmp_path):
    """bench.py's N > 1 path end to end, as the driver launches it (torch.distributed.run, one process per rank), on the
    GPUs that exist: two ranks share cuda:0 under gloo (GNET_BENCH_BACKEND).  Checks the JSON contract of the line rank 0
    prints: whole-job value, weak scaling, every rank's edge count and own ms/step, LPT image assignment."""
    import json, subprocess, sys, os, socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = str(s_.getsockname()[1]); s_.close()
    env = dict(os.environ, GNET_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", port, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--images", "2", "--dets", "500", "--blocks", "2", "--no-kernel-timing"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 3 and line["unit"] == "detections/sec"
    assert line["config"]["parallelism"] == "dp2" and line["config"]["image_assignment"].startswith("longest-processing-time")
    assert len(line["per_rank"]) == 2 and all(r["dets"] == 1000 and r["edges"] > 0 and r["ms_per_step"] > 0 for r in line["per_rank"])
    d = line["distributed"]
    assert d["backend"] == "gloo" and d["world_size"] == 2 and d["cuda_device_count"] >= 1 and d["distinct_devices"] == 1
    assert d["allreduce_samples"] == 3 and d["allreduce_us_rank0"] > 0 and all(r["device_index"] == 0 for r in line["per_rank"])
    assert sum(r["edges"] for r in line["per_rank"]) == line["config"]["edges_per_step_all_gpus"]
    assert abs(line["value"] - 2000 * 3 / (line["ms_per_step"] * 3e-3)) <= 1e-3 * line["value"]
    assert line["cpu_baseline"] is None and line["roofline"] is None          # rank 0 at N = 1 only / kernel timing off


def test_bench_eight_ranks_functional(tmp_path):
    """BASELINE configs[4]'s SHAPE before an 8-GPU node ever runs it: bench.py as the driver launches it with --gpus 8 -- eight
    processes (here sharing cuda:0 under gloo), a 64-image global step dealt by the longest-processing-time rule on edge counts,
    8 images per rank, one all-reduce per step.  Asserts 8 per-rank rows with equal image counts, edge balance max / mean <= 1.1,
    the summed gradient bitwise equal on every rank, and the whole-job value; then the RCCL log parser on eight synthetic rank
    files (the format RCCL writes with NCCL_DEBUG=INFO, NCCL_DEBUG_SUBSYS=INIT,GRAPH)."""
    import json, subprocess, sys, os, socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = str(s_.getsockname()[1]); s_.close()
    env = dict(os.environ, GNET_BENCH_BACKEND="gloo", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                          "--master-port", port, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                          "--images", "8", "--dets", "150", "--blocks", "2", "--no-kernel-timing"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    pr = line["per_rank"]
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["config"]["parallelism"] == "dp8"
    assert len(pr) == 8 and [r["rank"] for r in pr] == list(range(8))
    assert all(r["images"] == 8 and r["dets"] == 8 * 150 and r["ms_per_step"] > 0 for r in pr)
    edges = [r["edges"] for r in pr]
    assert sum(edges) == line["config"]["edges_per_step_all_gpus"]
    assert max(edges) <= 1.1 * (sum(edges) / 8.0), edges                    # LPT by edge count, equal image counts
    assert len({r["summed_grad_checksum"] for r in pr}) == 1, pr           # every replica holds the same summed gradient, bit for bit
    d = line["distributed"]
    assert d["backend"] == "gloo" and d["world_size"] == 8 and d["allreduce_samples"] == 2
    assert abs(line["value"] - 64 * 150 * 2 / (line["ms_per_step"] * 2e-3)) <= 1e-3 * line["value"]
    # the RCCL log parser over eight rank files
    sys.path.insert(0, root)
    import bench
    for r in range(8):
        (tmp_path / ("rccl_%d.log" % r)).write_text(
            "host:%d:%d [%d] NCCL INFO RCCL version 2.22.3+hip7.0 HEAD:abc\n"
            "host:%d:%d [%d] NCCL INFO comm 0x1 rank %d nranks 8 cudaDev %d busId %x000 commId 0xdead - Init START\n"
            "host:%d:%d [%d] NCCL INFO Channel 00/16 : 0 1 2 3 4 5 6 7\n"
            "host:%d:%d [%d] NCCL INFO comm 0x1 rank %d nranks 8 cudaDev %d busId %x000 - Init COMPLETE\n"
            % (100 + r, 200 + r, r, 100 + r, 200 + r, r, r, r, 5 + r, 100 + r, 200 + r, r, 100 + r, 200 + r, r, r, r, 5 + r))
    ev = bench.rccl_evidence(str(tmp_path / "rccl_%p.log"))
    assert ev["init_complete"] and "RCCL version" in ev["version"] and len(ev["bus_ids"]) == 8
    assert ev["rccl_ranks_seen"] == [{"rank": r, "nranks": 8} for r in range(8)] and ev["rings"]


def test_bench_one_rank_rccl_evidence_and_kernel_classes(tmp_path):
    """bench.py's N > 1 code path over RCCL with the one rank a one-GPU box has (GNET_BENCH_FORCE_DIST: a process group, the
    side-stream all-reduce launched without a wait, the next backward pass deferred behind it), and what the line then says
    about the run: RCCL's own INIT log parsed into distributed.rccl, the `graph` and `loss` kernel classes in the table."""
    import json, subprocess, sys, os, socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = str(s_.getsockname()[1]); s_.close()
    env = dict(os.environ, GNET_BENCH_FORCE_DIST="1")
    env.pop("GNET_BENCH_BACKEND", None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", port, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--images", "2", "--dets", "500", "--blocks", "2", "--no-other-configs", "--cpu-seconds", "0"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    d = line["distributed"]
    assert d["backend"] == "nccl" and d["world_size"] == 1 and d["allreduce_samples"] >= 3
    r = d["rccl"]
    assert r.get("init_complete") and {"rank": 0, "nranks": 1} in r["rccl_ranks_seen"] and "RCCL" in (r.get("version") or "")
    k = line["kernel_ms_per_step"]
    assert k.get("graph", 0) > 0 and k.get("loss", 0) > 0 and k.get("pw_fwd", 0) > 0 and k.get("edge_bwd", 0) > 0
    assert line["side_stream"]["graph_transpose_ms_alone"] > 0
    ab = line["side_stream"]["graph_transpose_placement_ms_per_step"]
    assert set(ab) == {"beside_pw_fwd", "after_forward"} and all(v > 0 for v in ab.values())


def test_checkpoint_round_trip_tf_bundle(tmp_path):
    """save(fmt="tf") writes a Saver V2 bundle keyed by the TF variable names (+ Adam slots, global_step); load() restores
    it into a fresh Gnet / Optimizer: same parameters, same next step."""
    from gossipnet_amd import checkpoint
    from gossipnet_amd.config import cfg
    from gossipnet_amd.train import Optimizer, train_step
    from gossipnet_amd.network import Gnet
    net, _ = make_pair(80, 2)
    opt = Optimizer(net)
    batch = make_image(80, 80, seed=0)
    for _ in range(2):
        train_step(net, opt, batch, 1e-3)
    path = checkpoint.save(net, checkpoint.checkpoint_name(2, str(tmp_path)), global_step=2, optimizer=opt, fmt="tf")
    assert checkpoint.latest_checkpoint(str(tmp_path)) == path
    net2 = Gnet(80)
    opt2 = Optimizer(net2)
    assert checkpoint.load(net2, path, optimizer=opt2) == 2
    assert torch.equal(net2.params, net.params) and torch.equal(opt2.m, opt.m) and torch.equal(opt2.v, opt.v)
    train_step(net, opt, batch, 1e-3)
    train_step(net2, opt2, batch, 1e-3)
    torch.cuda.synchronize()
    assert torch.equal(net2.params, net.params)


@pytest.mark.gpu
def test_a_diverged_run_yields_nan_not_a_memory_fault():
    """A diverging optimisation (SGD on the summed loss did it in a soak run) overflows the activations: the logits become
    inf / NaN.  The score ranking of the matching then compared NaNs (`>` and `==` false), left entries of the score order
    unwritten, and match_greedy followed stale indices of an earlier, larger batch out of bounds: a GPU memory fault instead of
    a NaN loss.  Non-finite parameters must give non-finite numbers and leave the device usable."""
    from gossipnet_amd.network import DeviceBatch
    net, _ = make_pair(80, 4)
    dev = torch.device("cuda", 0)
    big = DeviceBatch([make_image(n, 80, seed=s) for n, s in ((1163, 1), (1965, 2), (2123, 3), (439, 4))], dev)
    small = DeviceBatch([make_image(n, 80, seed=s) for n, s in ((300, 5), (1291, 6), (55, 7))], dev)
    net.run(big); torch.cuda.synchronize()                       # a larger batch first: its index arrays are what goes stale
    good = net.params.clone()
    net.run(small); torch.cuda.synchronize()
    ref_loss, ref_grads = net.loss.clone(), net.grads.clone()
    gen = torch.Generator(device="cpu").manual_seed(0)
    for scale in (1e4, 1e8, 1e16, 1e30, float("nan")):
        net.params.copy_(good * scale)
        if scale == 1e8:                                         # a mixed state: some tensors huge, some NaN, some sane
            m = torch.rand(good.numel(), generator=gen).to(dev)
            net.params.copy_(torch.where(m < 0.3, good * 1e20, torch.where(m < 0.4, torch.full_like(good, float("nan")), good)))
        net.run(small); torch.cuda.synchronize()
        assert net.det_gt_matching.min().item() >= -1 and net.det_gt_matching.max().item() < 400
        net.run(small, training=False); torch.cuda.synchronize()
    # the standalone op with NaN / inf scores: every detection still gets a label in {0, 1} and a weight
    from gossipnet_amd.matching_module import detection_matching
    iou = torch.rand(500, 40, generator=gen).to(dev)
    score = torch.randn(500, generator=gen).to(dev)
    score[::3] = float("nan"); score[1::7] = float("inf"); score[2::11] = -float("inf")
    labels, weights, assign = detection_matching(iou, score, torch.zeros(40, dtype=torch.bool, device=dev))
    torch.cuda.synchronize()
    assert set(np.unique(labels.cpu().numpy()).tolist()) <= {0.0, 1.0} and int(assign.max().item()) < 40
    # and the device still computes: the same step as before, bit for bit
    net.params.copy_(good)
    net.run(small); torch.cuda.synchronize()
    assert torch.equal(net.loss, ref_loss) and torch.equal(net.grads, ref_grads)


@pytest.mark.gpu
def test_a_rank_without_images_contributes_zero():
    """Fewer images than ranks in a data-parallel step: shard_images hands some rank an empty list.  Such a step must run --
    loss 0, gradient exactly 0 (also right after a step that left non-zero losses and gradients in the buffers) -- and so must
    an image without detections; the sum of the shards' gradients equals the gradient of the whole step."""
    from gossipnet_amd.network import DeviceBatch
    from gossipnet_amd.data_parallel import shard_images
    net, _ = make_pair(80, 2)
    dev = torch.device("cuda", 0)
    imgs = [make_image(120, 80, seed=1), make_image(75, 80, seed=2), make_image(33, 80, seed=3)]
    net.grad_scale = 1.0 / len(imgs)
    net.run(DeviceBatch(imgs, dev)); torch.cuda.synchronize()
    whole, whole_loss = net.grads.clone(), float(net.loss)
    assert whole.abs().max().item() > 0
    shards = [shard_images(imgs, r, 5, costs=[3.0, 2.0, 1.0]) for r in range(5)]
    assert sorted(len(s) for s in shards) == [0, 0, 1, 1, 1]
    total, total_loss = torch.zeros_like(whole), 0.0
    for s in shards:
        net.run(DeviceBatch(s, dev)); torch.cuda.synchronize()
        if not s:
            assert float(net.loss) == 0.0 and float(net.loss_normed) == 0.0 and net.grads.abs().max().item() == 0.0
        total += net.grads; total_loss += float(net.loss)
    assert abs(total_loss - whole_loss) <= 1e-5 * max(1.0, abs(whole_loss))
    assert float((total - whole).abs().max() / whole.abs().max()) <= 1e-5
    empty = {k: v[:0] for k, v in make_image(5, 80, seed=4).items()}
    net.run(DeviceBatch([empty], dev)); torch.cuda.synchronize()
    assert float(net.loss) == 0.0 and net.grads.abs().max().item() == 0.0


@pytest.mark.gpu
def test_backward_without_a_prepared_half_gives_the_same_bits():
    """gnet_backward(prepared = 0) builds the winner lists itself (gnet_backward_prepare phase 0 = 1 + 2 + 3 on its own stream); the
    product path runs the three phases on a side stream with two events (prepared = 1, prepared_event, positions_event).  Same
    gradients, bit for bit -- also with prepared = 1 and NO events after the caller has ordered the streams itself."""
    import ctypes as C
    import torch
    from gossipnet_amd import _lib
    from gossipnet_amd.config import experiment_cfg
    from gossipnet_amd.network import Gnet, DeviceBatch
    from gossipnet_amd.synthetic import make_image
    experiment_cfg()
    dev = torch.device("cuda", 0)
    net = Gnet(80, device=dev)
    batch = DeviceBatch([make_image(300, 80, seed=3), make_image(200, 80, seed=4)], dev)
    net.run(batch)
    torch.cuda.synchronize()
    ref = net.grads.clone()
    lib = net._lib
    vp = lambda t: C.c_void_p(t.data_ptr())
    s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for prepared in (0, 1):
        g2 = torch.full_like(ref, float("nan"))
        if prepared:      # the caller's own ordering: every phase on the launch stream, no events
            for phase in (1, 2, 3):
                _lib.check(lib.gnet_backward_prepare(C.byref(net._cfg), C.byref(net._shape), C.byref(net._inputs), vp(net.params),
                                                     C.byref(net._buf), phase, s), "gnet_backward_prepare")
        _lib.check(lib.gnet_backward(C.byref(net._cfg), C.byref(net._shape), C.byref(net._inputs), vp(net.params), C.byref(net._buf),
                                     vp(g2), prepared, None, None, s), "gnet_backward")
        torch.cuda.synchronize()
        assert torch.equal(g2, ref), "prepared = %d" % prepared
    # an unknown phase is refused
    assert lib.gnet_backward_prepare(C.byref(net._cfg), C.byref(net._shape), C.byref(net._inputs), vp(net.params),
                                     C.byref(net._buf), 7, s) != 0
