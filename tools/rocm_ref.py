#!/usr/bin/env python3
"""The reference's own execution path (torch.nn.functional conv1d / batch_norm / relu / dropout + autograd, restated in
oracle/torch_cpu_path.py) run by PyTorch-ROCm (MIOpen / rocBLAS) on this GPU: SURVEY 8(d)'s optional second comparator."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import torch_cpu_path as T  # noqa: E402
import videopose3d_amd as V  # noqa: E402

dev = "cuda:0"
fw, c, b = [3, 3, 3, 3, 3], 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.manual_seed(0)
m = V.TemporalModelOptimized1f(17, 2, 17, fw, dropout=0.25, channels=c)
sd = {k: v.detach().to(dev) for k, v in m.state_dict().items()}
x = (torch.randn(b, 243, 17, 2, device=dev) * 0.5).clamp(-1, 1)
tgt = torch.randn(b, 1, 17, 3, device=dev) * 0.3
t0 = time.perf_counter()
T.train_step(sd, x, tgt, fw, kind="strided", dropout=0.25)
torch.cuda.synchronize()
print("first training step (MIOpen find / compile included): %.1f s" % (time.perf_counter() - t0), flush=True)
for _ in range(2):
    T.train_step(sd, x, tgt, fw, kind="strided", dropout=0.25)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    T.train_step(sd, x, tgt, fw, kind="strided", dropout=0.25)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print("PyTorch-ROCm reference path, cfg3 train step B=%d: %.2f ms  %.0f frames/s" % (b, ms, b / ms * 1e3), flush=True)
with torch.no_grad():
    t0 = time.perf_counter()
    T.forward(sd, x, fw, kind="dilated", training=False)
    torch.cuda.synchronize()
    print("first eval forward: %.1f s" % (time.perf_counter() - t0), flush=True)
    T.forward(sd, x, fw, kind="dilated", training=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        T.forward(sd, x, fw, kind="dilated", training=False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
print("PyTorch-ROCm reference path, cfg2 eval forward B=%d: %.2f ms  %.0f frames/s" % (b, ms, b / ms * 1e3), flush=True)
