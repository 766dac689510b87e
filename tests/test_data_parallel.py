"""N > 1 path on CPU: world_size-2 gloo run of the data-parallel glue (shard -> local gradient -> ONE
all-reduce of the flat buffer) against the single-process result.  The local gradient is produced by
the CPU oracle here (there is no GPU); the HIP path plugs into exactly the same collective."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gossipnet_amd.data_parallel import allreduce_gradients, broadcast_parameters, shard_images
from gossipnet_amd.synthetic import make_image
from oracle import gnet_oracle as go

C_, B_, N_IMG = 1, 1, 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _images():
    return [make_image(12 + 3 * i, C_, seed=i) for i in range(N_IMG)]


def _local_grad(images, scale):
    orc = go.GnetOracle(C_, B_)
    total = None
    for im in images:
        _, g = orc.forward_backward(im)
        flat = go.flatten(g, C_, B_).astype(np.float64)
        total = flat if total is None else total + flat
    return torch.tensor(total * scale, dtype=torch.float32)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    params = torch.full((8,), float(rank))
    broadcast_parameters(params, dist, src=0)
    assert float(params.sum()) == 0.0                      # replicas start from rank 0's parameters
    mine = shard_images(_images(), rank, world)
    g = _local_grad(mine, 1.0 / N_IMG)                     # Gnet.grad_scale = 1 / (images of the global step)
    allreduce_gradients(g, dist)
    if rank == 0:
        np.save(out, g.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_equals_single_process(tmp_path):
    out = str(tmp_path / "g.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    ref = _local_grad(_images(), 1.0 / N_IMG).numpy()      # mean over the 4 images
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()


def test_shard_images_lpt_balances_edge_counts():
    imgs = list(range(8))
    costs = [100, 10, 10, 10, 90, 20, 20, 20]
    shards = [shard_images(imgs, r, 2, costs) for r in range(2)]
    assert sorted(shards[0] + shards[1]) == imgs
    loads = [sum(costs[i] for i in s) for s in shards]
    assert abs(loads[0] - loads[1]) <= 20
    assert shard_images(imgs, 1, 4) == [1, 5]


def test_rccl_log_parser(tmp_path):
    """bench.py's reading of RCCL's own INIT log (NCCL_DEBUG=INFO written to NCCL_DEBUG_FILE): ranks, version, bus ids, rings."""
    import bench
    p = tmp_path / "gnet_rccl_1.77.log"
    p.write_text("node:77:77 [0] NCCL INFO RCCL version : 2.26.6-HEAD:64f48b6\n"
                 "node:77:90 [0] NCCL INFO comm 0x1 rank 3 nranks 8 cudaDev 3 nvmlDev 3 busId dc000 commId 0x2 - Init START\n"
                 "node:77:90 [0] NCCL INFO Channel 00/16 : 0 1 2 3 4 5 6 7\n"
                 "node:77:90 [0] NCCL INFO Trees [0] 4/-1/-1->3->2\n"
                 "node:77:90 [0] NCCL INFO comm 0x1 rank 3 nranks 8 cudaDev 3 nvmlDev 3 busId dc000 commId 0x2 - Init COMPLETE\n")
    r = bench.rccl_evidence(str(tmp_path / "gnet_rccl_1.%p.log"))
    assert r["rccl_ranks_seen"] == [{"rank": 3, "nranks": 8}] and r["init_complete"] and r["bus_ids"] == ["busId dc000"]
    assert r["version"].startswith("RCCL version") and len(r["rings"]) == 2
    assert "note" in bench.rccl_evidence(None) and "note" in bench.rccl_evidence(str(tmp_path / "none.%p.log"))


def test_bench_prices_the_two_mfma_pipes():
    """bench.py's accounting of MFMA work: fp32 MFMAs against 157.3 TFLOP/s, bf16 MFMAs against 2516.6, and the FLOPs of a kernel's
    fp32 formulation beside them (edge_fwd_w: 72 bf16 MFMAs of 32x32x16 stand for 96 fp32 MFMAs of 32x32x2 per 32-edge tile)."""
    import bench
    E, N, W, R = 32 * 1000, 2000, 32 * 200, 32 * 700
    f32, bf16 = bench.executed_mfma_flops("edge_fwd", E, N, W, R)
    assert f32 == 0.0 and bf16 == 72 * 32768.0 * 1000
    assert bench.fp32_equivalent_flops("edge_fwd", E, N, W, R) == 96 * 4096.0 * 1000 == 12288.0 * E
    f32, bf16 = bench.executed_mfma_flops("edge_bwd", E, N, W, R)
    assert (f32, bf16) == (64 * 4096.0 * 200, 72 * 32768.0 * 200)
    assert bench.fp32_equivalent_flops("edge_bwd", E, N, W, R) == 160 * 4096.0 * 200
    # round 6: the pw-MLP kernels on the bf16 pipe -- pw_bwd_bf 216 bf16 MFMAs per wave and 32 listed rows (8 waves) for the 294 912
    # FLOPs per row of its four fp32 GEMMs; pw_fwd3 109 bf16 + 4 fp32 MFMAs per wave and 32 edges for 151 552 FLOPs per edge
    f32, bf16 = bench.executed_mfma_flops("pw_bwd_main", E, N, W, R)
    assert f32 == 0.0 and bf16 == 8 * 216 * 32768.0 * 700 and bench.fp32_equivalent_flops("pw_bwd_main", E, N, W, R) == 294912.0 * R
    f32, bf16 = bench.executed_mfma_flops("pw_fwd", E, N, W, R)
    assert (f32, bf16) == (8 * 4 * 4096.0 * 1000, 8 * 109 * 32768.0 * 1000) and bench.fp32_equivalent_flops("pw_fwd", E, N, W, R) == 151552.0 * E
    # a pure fp32 kernel: its formulation IS what it issues
    f32, bf16 = bench.executed_mfma_flops("node_fwd", E, N, W, R)
    assert bf16 == 0.0 and bench.fp32_equivalent_flops("node_fwd", E, N, W, R) == f32
    # pipe time: one second of each pipe's peak is one second
    assert abs(bench.pipe_seconds((bench.FP32_MFMA_PEAK_TFLOPS * 1e12, 0.0)) - 1.0) < 1e-12
    assert abs(bench.pipe_seconds((0.0, bench.BF16_MFMA_PEAK_TFLOPS * 1e12)) - 1.0) < 1e-12
    # 72 bf16 MFMAs take 72 x 32 cycles where 96 fp32 MFMAs took 96 x 64: the peaks are 16 x apart per FLOP
    assert abs(bench.BF16_MFMA_PEAK_TFLOPS / bench.FP32_MFMA_PEAK_TFLOPS - 16.0) < 0.01
    assert bench.executed_mfma_flops("graph", E, N, W, R) is None and bench.fp32_equivalent_flops("graph", E, N, W, R) is None
