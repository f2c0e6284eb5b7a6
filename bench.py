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
  cpu_baseline  : the reference's CPU path (ATen/oneDNN through oracle/torch_cpu_path.py, kind "port") on this
                  host's cores, bounded sample of the same workload; mpjpe_vs_ref = HIP vs that path on the sample
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


PMC_JSON = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
FAMILY_KERNELS = {"tconv_fwd": "k_rows_gemm<true,", "tconv_dgrad": "k_rows_gemm<false,", "tconv_wgrad": "k_red_gemm<"}


def pmc_traffic(family):
    """HBM-side bytes per launch of a GEMM family from the committed rocprofv3 PMC passes of THIS command
    (profiles/r01_pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate runs, FETCH_SIZE doubled per the
    gfx950 correction of MI355X_MICROARCH.md; tools/pmc_traffic.py).  PMC counters cannot be read from inside the
    timed process, so the value is the committed measurement, averaged over the family's launches like `achieved`."""
    try:
        with open(PMC_JSON) as f:
            tab = json.load(f)["kernels"]
    except (OSError, ValueError, KeyError):
        return None, "no committed PMC profile found (%s)" % os.path.relpath(PMC_JSON, ROOT)
    key = FAMILY_KERNELS[family].replace(" ", "")
    tot = n = 0.0
    for name, t in tab.items():
        if key in name.replace(" ", ""):
            launches = max(t["launches_fetch_pass"], t["launches_write_pass"], 1)
            tot += t["hbm_bytes_per_launch"] * launches
            n += launches
    if not n:
        return None, "kernel family not present in %s" % os.path.relpath(PMC_JSON, ROOT)
    return tot / n, ("bytes per launch (L2<->fabric: FETCH_SIZE x2 + WRITE_SIZE, Infinity-Cache hits included), mean "
                     "over the %d launches of the family in profiles/r01_pmc_traffic.json" % int(n))


def usable_cores():
    """Logical CPUs this process may actually run on: min(cpu_count, affinity mask, cgroup cpu quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, int(q / int(f.read()))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(dev, budget_s=4.0):
    """The reference's CPU path timed on this host's cores, next to the GPU number.

    /root/reference does not exist on the GPU box, so the timed code is oracle/torch_cpu_path.py: the same
    ATen/oneDNN kernels the reference's nn.Conv1d/BatchNorm1d/ReLU modules + autograd run on a CPU, wired as
    model.py:187-197 (kind "port"; pinned to reference-generated goldens by tests/test_oracle_golden.py).
    Bounded sample: the batch is sized so that one fwd+bwd takes ~budget_s seconds (CPU throughput is B-linear).
    The same leg is the checker for the metric's "MPJPE vs ref": HIP output vs this CPU path on the sample."""
    from oracle import torch_cpu_path as T
    from videopose3d_amd import TemporalModelOptimized1f
    cores = usable_cores()
    torch.manual_seed(0)
    m = TemporalModelOptimized1f(17, 2, 17, FW, dropout=0.0, channels=C).to(dev).train()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    gen = torch.Generator().manual_seed(1234)

    def run(bsz, iters):
        x, tgt = synthetic_batch(bsz, gen)
        ts, out = [], None
        for _ in range(iters):
            sdi = {k: v.clone() for k, v in sd.items()}
            t0 = time.perf_counter()
            _, out, _ = T.train_step(sdi, x, tgt, FW, kind="strided")
            ts.append(time.perf_counter() - t0)
        return x, out, ts

    # Thread count: the box reports more logical CPUs than the container may use; oversubscribed oneDNN threads
    # spin on each other (measured: 256 threads -> 25 s for a batch 8 threads finish in 0.13 s).  Walk the thread
    # count up from 8 and keep the fastest (stop at the first regression) -- `cores` reports what was used.
    best_t, best_dt, n = None, None, 8
    while n <= cores or best_t is None:
        n = min(n, cores)
        torch.set_num_threads(n)
        _, _, ts = run(32, 2)
        if best_dt is not None and ts[-1] > best_dt:
            break
        best_t, best_dt = n, ts[-1]
        if n == cores:
            break
        n *= 2
    torch.set_num_threads(best_t)
    _, _, ts = run(32, 2)                                   # warm-up + calibration at the chosen thread count
    bsz = int(min(B, max(32, round(32 * budget_s / ts[-1] / 32) * 32)))
    x, y_cpu, ts = run(bsz, 4)
    dt = float(np.mean(ts[1:]))
    with torch.no_grad():
        y_gpu = m(x.to(dev)).cpu()
    err = float(mpjpe(y_gpu, y_cpu))
    try:
        with open("/proc/cpuinfo") as f:
            cpu_name = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "?")
    except OSError:
        cpu_name = "?"
    base = dict(value=bsz / dt, unit="frames/s", cores=int(torch.get_num_threads()), kind="port",
                sample="oracle/torch_cpu_path.py (ATen/oneDNN conv1d/batch_norm/relu + autograd = what the reference "
                       "runs on CPU), TemporalModelOptimized1f arc 3,3,3,3,3 C=1024 train fwd+bwd (dropout off), "
                       "B=%d x 243 frames, 3 timed iters after 1 warm-up, %.2f s/iter, %d threads (fastest of a 8,16,.. sweep); "
                       "host: %d logical CPUs (%d usable by this container), %s"
                       % (bsz, dt, best_t, os.cpu_count() or 0, cores, cpu_name))
    return base, dict(value=err, unit="mpjpe (output units = metres) between the HIP path and the CPU reference path",
                      sample_b=bsz, tolerance=1e-3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eval", action="store_true", help="skip the cfg2 eval-forward section")
    args = ap.parse_args()

    from videopose3d_amd import TemporalModel, TemporalModelOptimized1f, dp, ops
    from videopose3d_amd import loss as vloss
    # VP3D_DIST_BACKEND / VP3D_BENCH_DEVICE are test hooks: "gloo" + device 0 let the N > 1 control flow (matched
    # collectives on every rank, no deadlock) be exercised on a box with a single GPU; the driver never sets them.
    rank, world, local = dp.init_from_env(os.environ.get("VP3D_DIST_BACKEND", "nccl"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    local = int(os.environ.get("VP3D_BENCH_DEVICE", local))
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
        loss = vloss.mpjpe(model(x), tgt)               # loss.py:11-17 on the HIP path (one kernel: value + gradient)
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

    # Everything below that runs training steps is executed by EVERY rank: a step contains collectives (the bucketed
    # all-reduces launched from inside backward and the final wait), so a rank-0-only step would deadlock for N > 1.
    if True:
        # ---- per-kernel-family roofline, measured live with HIP events on the launch stream -------------
        recs = []
        ops.set_profiler(recs)
        n_prof = 3
        for _ in range(n_prof):
            step()
        torch.cuda.synchronize()
        ops.set_profiler(None)
        fam = {}
        for name, flops, e0, e1, nbytes in recs:
            f = fam.setdefault(name, dict(flops=0.0, ms=0.0, calls=0, bytes=0.0))
            f["flops"] += flops
            f["bytes"] += nbytes
            f["ms"] += e0.elapsed_time(e1)
            f["calls"] += 1
        kernels = {k: dict(calls_per_step=v["calls"] // n_prof, ms_per_step=v["ms"] / n_prof,
                           avg_launch_ms=v["ms"] / v["calls"], tflops=v["flops"] / v["ms"] / 1e9,
                           algorithmic_bytes_per_launch=v["bytes"] / v["calls"])
                   for k, v in fam.items()}
        dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
        kname = {"tconv_fwd": "k_rows_gemm<true,*> (vp3d_tconv_fwd)", "tconv_dgrad": "k_rows_gemm<false,*> (vp3d_tconv_dgrad)",
                 "tconv_wgrad": "k_red_gemm<*> (vp3d_tconv_wgrad)"}[dom]
        traffic, traffic_note = pmc_traffic(dom)
        out["roofline"] = {"bound": "mfma", "achieved": kernels[dom]["tflops"], "peak": PEAK_F32_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": kernels[dom]["tflops"] / PEAK_F32_MFMA_TFLOPS,
                           "traffic": traffic, "traffic_note": traffic_note,
                           "algorithmic_bytes": kernels[dom]["algorithmic_bytes_per_launch"],
                           "kernel": kname, "launches_per_step": kernels[dom]["calls_per_step"],
                           "avg_launch_ms": kernels[dom]["avg_launch_ms"],
                           "note": "achieved = sum of algorithmic conv FLOPs (2*M*N*K) of the family's launches / "
                                   "sum of their HIP-event durations on the launch stream"}
        out["kernels"] = kernels
        gemm_ms = sum(v["ms_per_step"] for v in kernels.values())
        out["non_gemm_ms_per_step"] = ms_per_step - gemm_ms if world == 1 else None

        # ---- the callers either side of the stack, outside the metric -------------------------
        def timed(fn, n=5):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

        opt = torch.optim.Adam(model.parameters(), lr=1e-3, amsgrad=True)      # run.py:252,264
        out["adam_torch_ms"] = timed(opt.step)
        del opt
        from videopose3d_amd.optim import FlatAdam
        from videopose3d_amd.generators import ChunkedGenerator
        fopt = FlatAdam(model.parameters(), lr=1e-3, amsgrad=True, grad_sync=sync)
        out["adam_ms"] = timed(fopt.step)                                       # vp3d_adam_step, one pass
        rng = np.random.RandomState(0)
        lens = [3000 + 517 * i for i in range(24)]                              # ~216k frames of synthetic "videos"
        p2 = [rng.standard_normal((n, 17, 2)).astype(np.float32) for n in lens]
        p3 = [rng.standard_normal((n, 17, 3)).astype(np.float32) for n in lens]
        gen_dev = ChunkedGenerator(B, None, p3, p2, 1, pad=(RF - 1) // 2, shuffle=True, augment=True,
                                   kps_left=[1, 3, 5, 7, 9, 11, 13, 15], kps_right=[2, 4, 6, 8, 10, 12, 14, 16],
                                   joints_left=[4, 5, 6, 11, 12, 13], joints_right=[1, 2, 3, 14, 15, 16], device=dev)
        it = gen_dev.next_epoch()
        out["batch_gather_ms"] = timed(lambda: next(it), n=10)                  # vp3d_gather_chunks, B=1024 x 243 frames

        def full_step():
            _, b3, b2 = next(it)
            b3[:, :, 0] = 0                                                     # run.py:407
            fopt.zero_grad()
            vloss.mpjpe(model(b2), b3).backward()
            sync.sync()                                                         # no-op for one GPU
            fopt.step()
        ms_full = timed(full_step, n=10)
        out["full_step"] = {"what": "device batch assembly + fwd + bwd + fused Adam (run.py:401-420 end to end), B=1024",
                            "ms": ms_full, "frames_per_s": B / ms_full * 1e3}
        del fopt, gen_dev, it

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
        big = [(f, e0.elapsed_time(e1)) for _, f, e0, e1, _b in recs if f > 1e11]
        tf = FLOP_EVAL_PER_FRAME * B / dte / 1e12
        out["cfg2_eval_fwd"] = {"workload": "TemporalModel eval forward (BN folded), arc 3,3,3,3,3 C=1024 B=1024 T=243",
                                "ms": dte * 1e3, "frames_per_s": B / dte, "tflops": tf,
                                "frac_of_mfma_peak": tf / PEAK_F32_MFMA_TFLOPS, "iters": k_eval,
                                "big_gemm_tflops": [round(f / ms / 1e9, 1) for f, ms in big]}
        del ev
        torch.cuda.empty_cache()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"], out["mpjpe_vs_ref"] = cpu_baseline(dev)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
