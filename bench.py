#!/usr/bin/env python
"""bench.py -- Gnet forward+backward throughput on synthetic N-detection images (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of `--images` synthetic images per GPU
(default 8 = BASELINE config 5's per-GPU share: 64 images/step on 8 GPUs), N=2000 detections,
80 classes, 16 blocks, fp32: graph build (IoU sweep + ordered CSR) -> pairwise features + pw-MLP ->
16 blocks -> head -> det_anno_iou + detection matching + weighted sigmoid x-ent -> full backward
(all parameter gradients) [-> one RCCL all-reduce of the flat gradient when N > 1].
Inputs are resident in HBM before the timed region.  value = detections/sec over all ranks.

The JSON line also carries
  roofline     : the dominant kernel class (by HIP-event time inside the timed region), its
                 algorithmic FLOPs per launch (SURVEY 8d formulas) / its average launch duration,
                 against the fp32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md)
  cpu_baseline : the CPU oracle (oracle/gnet_oracle.py, a port of the reference TF-CPU path)
                 timed on this host on a bounded sample (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
HBM_PEAK_GBS = 8000.0


def algorithmic_flops(cls, E, N, C):
    """Nominal (reference-algorithm) FLOPs of one launch of kernel class `cls` (2 per MAC)."""
    dpw = 2 * (C if C > 1 else 1) + 7
    table = {
        "pw_fwd": 2.0 * E * (256 * dpw + 256 * 256 + 256 * 32),
        "edge_fwd": 2.0 * E * (96 * 64 + 64 * 64),
        "node_fwd": 2.0 * N * (128 * 32 + 64 * 64 + 64 * 128),
        "edge_bwd": 4.0 * E * (96 * 64 + 64 * 64),
        "pw_bwd_main": 4.0 * E * (256 * 256 + 256 * 32),
        "pw_bwd_w1": 2.0 * E * dpw * 256,
        "blk_bwd_post": 4.0 * N * (64 * 64 + 64 * 128),
        "blk_bwd_pre": 4.0 * N * (128 * 32),
        "head_bwd": 4.0 * N * (2 * 128 * 128 + 128),
    }
    return table.get(cls)


def step_flops(E, N, C, B=16):
    """fwd+bwd nominal FLOPs of one image batch (1 596 416 E + 1 770 240 N for C=80, B=16)."""
    dpw = 2 * (C if C > 1 else 1) + 7
    pw = 2.0 * E * (256 * dpw + 256 * 256 + 256 * 32)
    fwd = pw + B * (2.0 * E * (96 * 64 + 64 * 64) + 2.0 * N * (128 * 32 + 64 * 64 + 64 * 128)) + 2.0 * N * (2 * 128 * 128 + 128)
    return 3.0 * fwd - 2.0 * E * dpw * 256


def cpu_baseline(images, num_classes, num_blocks, budget_s):
    from oracle import gnet_oracle as go
    orc = go.GnetOracle(num_classes, num_blocks)
    n_done, dets, t_total = 0, 0, 0.0
    for im in images:
        t0 = time.perf_counter()
        orc.forward_backward(im)
        dt = time.perf_counter() - t0
        t_total += dt
        n_done += 1
        dets += im["dets"].shape[0]
        if t_total > budget_s:
            break
    return {"value": dets / t_total, "unit": "detections/sec", "cores": int(torch.get_num_threads()),
            "kind": "port", "sample": "%d image(s) of the same workload, fwd+bwd, torch-CPU fp32 oracle, %.1f s" % (n_done, t_total)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--images", type=int, default=8, help="images per step per GPU")
    ap.add_argument("--dets", type=int, default=2000)
    ap.add_argument("--classes", type=int, default=80)
    ap.add_argument("--blocks", type=int, default=16)
    ap.add_argument("--preset", default="dense", choices=["dense", "coco_like"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline (0 = skip)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--lanes", type=int, default=1, help="split the step's images over this many streams "
                    "(replica Gnets sharing the variables): overlaps MFMA-bound and HBM-bound kernels")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group("nccl", device_id=dev)
        dist = dist_mod

    from gossipnet_amd.config import cfg, reset_cfg
    from gossipnet_amd.network import Gnet, DeviceBatch
    from gossipnet_amd.synthetic import make_image
    from gossipnet_amd.data_parallel import allreduce_gradients
    reset_cfg()
    cfg.gnet.num_blocks = args.blocks
    net = Gnet(args.classes, device=dev)
    nets = [net] + [Gnet(args.classes, device=dev, reuse=True) for _ in range(args.lanes - 1)]
    for n_ in nets:   # gradient of the mean over all images of the global batch
        n_.grad_scale = 1.0 / (args.images * world)

    images = [make_image(args.dets, args.classes, seed=rank * args.images + i, preset=args.preset) for i in range(args.images)]
    # inputs resident in HBM before the timed region
    batches = [DeviceBatch(images[l::args.lanes], dev) for l in range(args.lanes)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(args.lanes)] if args.lanes > 1 else [None]

    def step():
        if args.lanes == 1:
            net.run(batches[0])
        else:
            main_s = torch.cuda.current_stream(dev)
            for n_, b_, s_ in zip(nets, batches, streams):
                s_.wait_stream(main_s)
                with torch.cuda.stream(s_):
                    n_.begin(b_)
            for n_, s_ in zip(nets, streams):
                with torch.cuda.stream(s_):
                    n_.run()
            for n_, s_ in zip(nets[1:], streams[1:]):
                main_s.wait_stream(s_)
            main_s.wait_stream(streams[0])
            for n_ in nets[1:]:
                net.grads.add_(n_.grads)
        if dist is not None:
            allreduce_gradients(net.grads, dist)

    for _ in range(args.warmup):
        step()
    E = sum(int(n_.num_edges) for n_ in nets)
    # Per-kernel HIP events cost ~4% of the step when every launch is bracketed (~230 launches), so the timed
    # region brackets only the DOMINANT kernel class (picked from one fully instrumented, untimed step);
    # the complete per-class table is measured in a second, untimed pass after the timed region.
    dominant = None
    if not args.no_kernel_timing:
        for n_ in nets:
            n_.enable_kernel_timing(classes=None, capacity=512)
        step()
        torch.cuda.synchronize()
        probe = {}
        for n_ in nets:
            for k_, (ms_, c_) in n_.read_kernel_timing().items():
                probe[k_] = probe.get(k_, 0.0) + ms_
        dominant = max(probe.items(), key=lambda kv: kv[1])[0]
        for n_ in nets:
            n_.enable_kernel_timing(classes=[dominant], capacity=(args.steps + 1) * 64)

    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        et = torch.tensor([E], dtype=torch.float64, device=dev)
        dist.all_reduce(et)
        e_total = float(et.item())
    else:
        e_total = float(E)

    dets_per_step = args.dets * args.images * world
    value = dets_per_step * args.steps / elapsed

    roofline = None
    timing = {}
    table = {}
    if not args.no_kernel_timing:
        for n_ in nets:
            for k_, (ms_, c_) in n_.read_kernel_timing().items():
                pm_, pc_ = timing.get(k_, (0.0, 0))
                timing[k_] = (pm_ + ms_, pc_ + c_)
        # second pass (untimed): every class, for the kernel_ms_per_step table
        table_steps = min(args.steps, 5)
        for n_ in nets:
            n_.enable_kernel_timing(classes=None, capacity=(table_steps + 1) * 256)
        for _ in range(table_steps):
            step()
        torch.cuda.synchronize()
        for n_ in nets:
            for k_, (ms_, c_) in n_.read_kernel_timing().items():
                table[k_] = table.get(k_, 0.0) + ms_ / table_steps
        N_local = args.dets * args.images
        dom = max(timing.items(), key=lambda kv: kv[1][0]) if timing else None
        if dom is not None:
            cls, (ms, cnt) = dom
            fl = algorithmic_flops(cls, E / args.lanes, N_local / args.lanes, args.classes)
            if fl is not None:
                avg_s = ms / cnt * 1e-3
                ach = fl / avg_s / 1e12
                traffic = None
                tf = os.path.join(ROOT, "profiles", "r01_traffic.json")
                if os.path.exists(tf) and args.dets == 2000 and args.images == 8 and args.preset == "dense" and args.classes == 80:
                    # HBM bytes per launch from a separate rocprofv3 --pmc run of this same workload
                    traffic = json.load(open(tf))["kernels"].get(cls, {}).get("hbm_bytes")
                roofline = {"bound": "mfma", "kernel": cls, "achieved": round(ach, 3), "peak": FP32_MFMA_PEAK_TFLOPS,
                            "unit": "TFLOP/s", "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                            "avg_launch_us": round(avg_s * 1e6, 2), "launches": cnt,
                            "flops_per_launch": fl}

    if rank == 0:
        out = {
            "metric": "detections/sec Gnet fwd+bwd, N=%d/%d-class" % (args.dets, args.classes),
            "value": round(value, 1), "unit": "detections/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2] coco_multiclass 80-way N=2000 (synthetic '%s' preset), "
                                   "%d images/step/GPU (configs[4] per-GPU share), %d blocks" % (args.preset, args.images, args.blocks),
                       "dets_per_image": args.dets, "images_per_step_per_gpu": args.images, "num_classes": args.classes,
                       "num_blocks": args.blocks, "edges_per_step_all_gpus": e_total,
                       "edges_per_det": round(e_total / dets_per_step, 2), "parallelism": "dp%d" % world, "lanes_per_gpu": args.lanes,
                       "step": "graph build + fwd + matching/loss + bwd" + (" + RCCL all-reduce" if world > 1 else "")},
            "whole_step": {"note": "nominal = FLOPs of the reference's dense algorithm (SURVEY 8d); the sparse SegmentMax backward executes fewer",
                           "nominal_tflops": round(step_flops(E, args.dets * args.images, args.classes, args.blocks) * world * args.steps / elapsed / 1e12, 3),
                           "frac_fp32_mfma_peak": round(step_flops(E, args.dets * args.images, args.classes, args.blocks) * args.steps / elapsed / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)},
            "roofline": roofline,
            "kernel_ms_per_step": {k: round(v, 4) for k, v in sorted(table.items(), key=lambda kv: -kv[1])},
        }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(images, args.classes, args.blocks, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
