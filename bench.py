#!/usr/bin/env python3
"""Benchmark of the VideoPose3D temporal-model hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json `metric`: "frames/sec (fwd+bwd) 243-frame arc=3,3,3,3,3 B=1024"):
  one step = one training pass (forward + backward incl. BatchNorm statistics and dropout p=0.25) of
  TemporalModelOptimized1f, arc 3,3,3,3,3, C=1024, over one synthetic batch of B=1024 RF-length (243-frame)
  windows that is already resident in HBM; for N>1 one such batch PER RANK (weak scaling, global batch
  N*1024) plus the flat-gradient sum all-reduce over RCCL.  frames/s = predicted frames (= batch elements) per
  second, whole job.  The optimizer is torch.optim.Adam in run.py (out of scope) and is NOT in the timed
  region; its cost is reported separately (`adam_ms`).
Extra fields in the same JSON line:
  cfg2_eval_fwd : BASELINE.json configs[1] -- TemporalModel eval forward, B=1024, T=243 (5.34 TFLOP / call)
  roofline      : dominant GEMM kernel family of the step, algorithmic FLOPs / HIP-event launch durations
  cpu_baseline  : the numpy oracle ("port") on this host's cores, bounded sample of the same workload
Rank 0 prints ONE JSON line on stdout.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FW = [3, 3, 3, 3, 3]
C = 1024
B = 1024
RF = 243
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 4 SIMD x 64 FLOP/clk x 2.4 GHz
FLOP_TRAIN_PER_FRAME = 1023866880     # fwd 352,569,344 + bwd 671,297,536 (conv MACs x 2)
FLOP_EVAL_PER_FRAME = 5217830912      # TemporalModel forward on a 243-frame window


def synthetic_batch(batch, gen):
    x = (torch.randn(batch, RF, 17, 2, generator=gen) * 0.5).clamp(-1, 1)      # screen-normalised keypoints
    tgt = torch.randn(batch, 1, 17, 3, generator=gen) * 0.3
    tgt[:, :, 0] = 0                                                            # run.py:407
    return x, tgt


def mpjpe(pred, target):                                                        # loss.py:11-17
    return torch.mean(torch.norm(pred - target, dim=len(target.shape) - 1))


def cpu_baseline(sample_b=48, iters=2):
    """The oracle (kind "port": numpy restatement of the reference's algorithm) timed on the host cores."""
    from oracle import temporal_oracle as O
    from videopose3d_amd import TemporalModelOptimized1f
    try:
        from threadpoolctl import threadpool_info
        threads = max([i.get("num_threads", 1) for i in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    torch.manual_seed(0)
    m = TemporalModelOptimized1f(17, 2, 17, FW, dropout=0.0, channels=C)
    sd = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    gen = torch.Generator().manual_seed(1234)
    x, tgt = synthetic_batch(sample_b, gen)
    x, tgt = x.numpy(), tgt.numpy()
    times = []
    for i in range(iters + 1):
        t0 = time.perf_counter()
        y, cache, _ = O.forward(sd, x, FW, kind="strided", training=True)
        O.backward(cache, O.mpjpe_grad(y, tgt))
        if i:
            times.append(time.perf_counter() - t0)
    dt = float(np.mean(times))
    return dict(value=sample_b / dt, unit="frames/s", cores=int(threads), kind="port",
                sample="numpy oracle, TemporalModelOptimized1f arc 3,3,3,3,3 C=1024 fwd+bwd (dropout off), "
                       "B=%d, %d timed iters after 1 warm-up, %.2f s/iter; host has %d logical cores"
                       % (sample_b, iters, dt, os.cpu_count() or 0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eval", action="store_true", help="skip the cfg2 eval-forward section")
    args = ap.parse_args()

    from videopose3d_amd import TemporalModel, TemporalModelOptimized1f, dp, ops
    rank, world, local = dp.init_from_env("nccl")
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    torch.manual_seed(0)
    model = TemporalModelOptimized1f(17, 2, 17, FW, causal=False, dropout=0.25, channels=C).to(dev).train()
    sync = dp.FlatGradSync(model.parameters(), world=world, direct_module=model)
    sync.broadcast_parameters(model.buffers())
    gen = torch.Generator().manual_seed(1234 + rank)
    x, tgt = synthetic_batch(B, gen)
    x, tgt = x.to(dev), tgt.to(dev)                      # inputs resident in HBM before the timed region

    def step():
        sync.zero_grad()
        loss = mpjpe(model(x), tgt)
        loss.backward()
        sync.sync()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt

    out = {
        "metric": "frames/sec (fwd+bwd) 243-frame arc=3,3,3,3,3 B=1024", "value": value, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "cfg3: TemporalModelOptimized1f train step fwd+bwd (BN stats, dropout 0.25), "
                               "arc 3,3,3,3,3 C=1024, per-GPU B=1024 x 243 frames x 17 joints",
                   "global_batch": world * B, "parallelism": "dp%d" % world,
                   "grad_allreduce_bytes": sync.numel * 4 if world > 1 else 0},
        "step_tflops": FLOP_TRAIN_PER_FRAME * value / 1e12,
        "step_frac_of_mfma_peak": FLOP_TRAIN_PER_FRAME * value / 1e12 / (PEAK_F32_MFMA_TFLOPS * world),
    }

    if rank == 0:
        # ---- per-kernel-family roofline, measured live with HIP events on the launch stream -------------
        recs = []
        ops.set_profiler(recs)
        n_prof = 3
        for _ in range(n_prof):
            step()
        torch.cuda.synchronize()
        ops.set_profiler(None)
        fam = {}
        for name, flops, e0, e1 in recs:
            f = fam.setdefault(name, dict(flops=0.0, ms=0.0, calls=0))
            f["flops"] += flops
            f["ms"] += e0.elapsed_time(e1)
            f["calls"] += 1
        kernels = {k: dict(calls_per_step=v["calls"] // n_prof, ms_per_step=v["ms"] / n_prof,
                           avg_launch_ms=v["ms"] / v["calls"], tflops=v["flops"] / v["ms"] / 1e9)
                   for k, v in fam.items()}
        dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
        kname = {"tconv_fwd": "k_rows_gemm<true,*> (vp3d_tconv_fwd)", "tconv_dgrad": "k_rows_gemm<false,*> (vp3d_tconv_dgrad)",
                 "tconv_wgrad": "k_red_gemm<*> (vp3d_tconv_wgrad)"}[dom]
        out["roofline"] = {"bound": "mfma", "achieved": kernels[dom]["tflops"], "peak": PEAK_F32_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": kernels[dom]["tflops"] / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                           "kernel": kname, "launches_per_step": kernels[dom]["calls_per_step"],
                           "avg_launch_ms": kernels[dom]["avg_launch_ms"],
                           "note": "achieved = sum of algorithmic conv FLOPs (2*M*N*K) of the family's launches / "
                                   "sum of their HIP-event durations on the launch stream"}
        out["kernels"] = kernels
        gemm_ms = sum(v["ms_per_step"] for v in kernels.values())
        out["non_gemm_ms_per_step"] = ms_per_step - gemm_ms if world == 1 else None

        # ---- optimizer cost, outside the metric ---------------------------------------------------------
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, amsgrad=True)      # run.py:252,264
        opt.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            opt.step()
        torch.cuda.synchronize()
        out["adam_ms"] = (time.perf_counter() - t0) / 5 * 1e3
        del opt

    del model, sync
    torch.cuda.empty_cache()

    if rank == 0 and not args.no_eval:
        # ---- BASELINE.json configs[1]: dense eval forward, B=1024, T=243 ------------------------------
        torch.manual_seed(0)
        ev = TemporalModel(17, 2, 17, FW, channels=C).to(dev).eval()
        k_eval = max(3, args.steps // 4)
        with torch.no_grad():
            for _ in range(2):
                ev(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k_eval):
                ev(x)
            torch.cuda.synchronize()
            dte = (time.perf_counter() - t0) / k_eval
            recs = []
            ops.set_profiler(recs)
            ev(x)
            torch.cuda.synchronize()
            ops.set_profiler(None)
        big = [(f, e0.elapsed_time(e1)) for _, f, e0, e1 in recs if f > 1e11]
        tf = FLOP_EVAL_PER_FRAME * B / dte / 1e12
        out["cfg2_eval_fwd"] = {"workload": "TemporalModel eval forward (BN folded), arc 3,3,3,3,3 C=1024 B=1024 T=243",
                                "ms": dte * 1e3, "frames_per_s": B / dte, "tflops": tf,
                                "frac_of_mfma_peak": tf / PEAK_F32_MFMA_TFLOPS, "iters": k_eval,
                                "big_gemm_tflops": [round(f / ms / 1e9, 1) for f, ms in big]}
        del ev
        torch.cuda.empty_cache()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
