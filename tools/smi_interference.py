#!/usr/bin/env python3
"""Does a `rocm-smi` poll on the same box disturb the timed step?  (BENCH_r02: the driver samples GPU use every ~5 s; its
0.12 s headline window measured 5.997 ms / step where the same process later ran 4.76 ms.)

Runs the cfg3 step for `seconds` with a HIP event between the steps, first alone, then with `rocm-smi --showuse` started every
`period` seconds in the background, and prints the distribution of the per-step GPU times and the largest 20-step-window mean
of both phases.  Also the FIRST windows of the process (cold clocks / allocator) step by step."""
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402
from videopose3d_amd import dp, loss as vloss  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
period = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
dev = "cuda:0"
torch.manual_seed(0)
x = (torch.randn(1024, 243, 17, 2, device=dev) * 0.5).clamp(-1, 1)
tgt = torch.randn(1024, 1, 17, 3, device=dev) * 0.3
m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3, 3, 3], dropout=0.25, channels=1024).to(dev).train()
sync = dp.FlatGradSync(m.parameters(), world=1, direct_module=m)


def step():
    sync.zero_grad()
    vloss.mpjpe(m(x), tgt).backward()
    sync.sync()


def run(n):
    """n steps in windows of 20 (fence between windows, as bench.py): per-step event times + per-window wall / host times"""
    per, wall, host, mem = [], [], [], []
    for w0 in range(0, n, 20):
        k = min(20, n - w0)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            evs[i].record()
            step()
        evs[k].record()
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        wall.append((time.perf_counter() - t0) / k * 1e3)
        host.append(th / k * 1e3)
        mem.append(torch.cuda.memory_reserved() / 2 ** 30)
        per += [evs[i].elapsed_time(evs[i + 1]) for i in range(k)]
    return per, wall, host, mem


def describe(tag, per, wall, host, mem):
    s = sorted(per)
    print("%s: %d steps  median %.3f  p90 %.3f  p99 %.3f  max %.3f ms;  windows (wall ms/step): %s" %
          (tag, len(per), s[len(s) // 2], s[int(len(s) * 0.9)], s[int(len(s) * 0.99)], s[-1],
           " ".join("%.3f" % v for v in wall)), flush=True)
    print("    host enqueue ms/step per window: %s" % " ".join("%.2f" % v for v in host), flush=True)
    print("    reserved GiB after each window:  %s" % " ".join("%.2f" % v for v in mem), flush=True)


# ---- cold process: 5 warm-up steps like the driver's command, then the first windows, step by step -----------------------
for _ in range(5):
    step()
per, wall, host, mem = run(100)
print("first 40 steps after 5 warm-up steps (GPU ms): " + " ".join("%.2f" % v for v in per[:40]), flush=True)
describe("cold process, first 100 steps", per, wall, host, mem)
n = int(seconds / (wall[-1] / 1e3))
describe("alone", *run(n))

stop = False


def poll():
    while not stop:
        t0 = time.perf_counter()
        try:
            subprocess.run(["rocm-smi", "--showuse", "--showpower"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=20)
        except Exception as e:  # noqa: BLE001
            print("rocm-smi failed:", e, flush=True)
            return
        poll.took.append(time.perf_counter() - t0)
        time.sleep(period)


poll.took = []
th = threading.Thread(target=poll, daemon=True)
th.start()
describe("with rocm-smi every %.1f s" % period, *run(n))
stop = True
th.join(timeout=30)
if poll.took:
    print("rocm-smi calls: %d, %.2f s each" % (len(poll.took), sum(poll.took) / len(poll.took)), flush=True)
describe("alone again", *run(n))
