#!/usr/bin/env python3
"""f16x3 (split-fp16) vs f32 math on whole models: parity of outputs / running stats / gradients, then timing."""
import copy
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402
from videopose3d_amd import engine  # noqa: E402

engine.S16_MIN_FORWARD_FLOPS.update({True: 0.0, False: 0.0})   # compare the engines also on the small models

dev = "cuda:0"


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def parity(fw, c, b, p):
    torch.manual_seed(0)
    m32 = V.TemporalModelOptimized1f(17, 2, 17, fw, dropout=p, channels=c).to(dev).train()
    m32.math = "f32"
    m16 = copy.deepcopy(m32)
    m16.math = "f16x3"
    for m in (m32, m16):
        m._drop_seed, m._drop_calls = 1234567, 0
    rf = m32.receptive_field()
    x = (torch.randn(b, rf, 17, 2, device=dev) * 0.5).clamp(-1, 1)
    tgt = torch.randn(b, 1, 17, 3, device=dev) * 0.3
    outs = []
    for m in (m32, m16):
        y = m(x)
        loss = torch.mean(torch.norm(y - tgt, dim=3))
        loss.backward()
        outs.append(y.detach())
    print("train fw=%s C=%d B=%d p=%.2f: out rel %.2e (mpjpe %.2e)" % (fw, c, b, p, rel(outs[1], outs[0]),
          float(torch.mean(torch.norm(outs[1] - outs[0], dim=3)))))
    worst = 0.0
    for (k, a), (_, q) in zip(m16.named_parameters(), m32.named_parameters()):
        r = rel(a.grad, q.grad)
        worst = max(worst, r)
        if r > 1e-4:
            print("   grad %-28s rel %.2e" % (k, r))
    print("   worst grad rel %.2e" % worst)
    for (k, a), (_, q) in zip(m16.named_buffers(), m32.named_buffers()):
        if a.dtype.is_floating_point and rel(a, q) > 1e-5:
            print("   buffer %-28s rel %.2e" % (k, rel(a, q)))
    # eval (dilated) parity with the trained-state buffers
    e32 = V.TemporalModel(17, 2, 17, fw, channels=c).to(dev).eval()
    e32.load_state_dict(m32.state_dict())
    e32.math = "f32"
    e16 = copy.deepcopy(e32)
    e16.math = "f16x3"
    xe = (torch.randn(4, rf + 20, 17, 2, device=dev) * 0.5).clamp(-1, 1)
    with torch.no_grad():
        a, q = e16(xe), e32(xe)
    print("   eval dilated: out rel %.2e (mpjpe %.2e)" % (rel(a, q), float(torch.mean(torch.norm(a - q, dim=3)))))


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def perf():
    fw, c, b = [3, 3, 3, 3, 3], 1024, 1024
    torch.manual_seed(0)
    x = (torch.randn(b, 243, 17, 2, device=dev) * 0.5).clamp(-1, 1)
    tgt = torch.randn(b, 1, 17, 3, device=dev) * 0.3
    for math in ("f32", "f16x3"):
        m = V.TemporalModelOptimized1f(17, 2, 17, fw, dropout=0.25, channels=c).to(dev).train()
        m.math = math

        def step():
            m.zero_grad(set_to_none=True)
            torch.mean(torch.norm(m(x) - tgt, dim=3)).backward()
        for _ in range(3):
            step()
        ms = timed(step, 10)
        print("cfg3 train step  math=%-6s %7.3f ms  %8.0f frames/s" % (math, ms, b / ms * 1e3), flush=True)
        del m
        e = V.TemporalModel(17, 2, 17, fw, channels=c).to(dev).eval()
        e.math = math
        with torch.no_grad():
            e(x)
            ms = timed(lambda: e(x), 4)
        print("cfg2 eval fwd    math=%-6s %7.3f ms  %8.0f frames/s" % (math, ms, b / ms * 1e3), flush=True)
        del e
        torch.cuda.empty_cache()


if __name__ == "__main__":
    parity([3, 3, 3], 128, 16, 0.0)
    parity([3, 3, 3], 128, 16, 0.25)
    parity([3, 3, 3, 3], 256, 8, 0.25)
    if "--noperf" not in sys.argv:
        perf()
