#!/usr/bin/env python3
"""Upper bound of what ANY fusion of the BatchNorm-backward column sums into another kernel could save (verdict r2 item 6):
the cfg3 training step with the eight `k_bn_bwd_reduce_strips` launches REMOVED -- their three small results (dgamma, dbeta,
bound of dy) come from a recorded step through one multi-tensor copy per layer, so the values every later kernel sees stay
realistic (bounds doubled: the dropout masks change from step to step).  Interleaved with the normal step in one process.
The numbers of the "removed" step are not a training step (gradients are stale); only its time means something."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402
from videopose3d_amd import _lib, dp, loss as vloss  # noqa: E402

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.manual_seed(0)
x = (torch.randn(B, 243, 17, 2, device=dev) * 0.5).clamp(-1, 1)
tgt = torch.randn(B, 1, 17, 3, device=dev) * 0.3
m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3, 3, 3], dropout=0.25, channels=1024).to(dev).train()
sync = dp.FlatGradSync(m.parameters(), world=1, direct_module=m)

L = _lib.lib()
real = L.vp3d_bn_bwd_reduce_fin_s16
mode = {"v": "real"}
replay = {}
calls = {"n": 0}


def _addr(a):
    return a if isinstance(a, int) else (a.value if hasattr(a, "value") else a)


def patched(stream, m_rows, c, go, y, mu, inv, bits, p, sc, go_bound, parts, gparts, tickets, dgam, dbet, dy_bound, *rest):
    if go is None or mode["v"] == "real":
        return real(stream, m_rows, c, go, y, mu, inv, bits, p, sc, go_bound, parts, gparts, tickets, dgam, dbet, dy_bound, *rest)
    key = (m_rows, _addr(dgam))
    if mode["v"] == "record":
        rc = real(stream, m_rows, c, go, y, mu, inv, bits, p, sc, go_bound, parts, gparts, tickets, dgam, dbet, dy_bound, *rest)
        replay[key] = [view(_addr(dgam), c).clone(), view(_addr(dbet), c).clone(), view(_addr(dy_bound), 32).clone() * 2.0]
        return rc
    calls["n"] += 1
    torch._foreach_copy_([view(_addr(dgam), c), view(_addr(dbet), c), view(_addr(dy_bound), 32)], replay[key])
    return 0


L.vp3d_bn_bwd_reduce_fin_s16 = patched


class _Cap:
    """Device memory at ptr as seen through the CUDA array interface (the flat gradient buffer / a bound array)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def view(ptr, n):
    return torch.as_tensor(_Cap(ptr, n), device=dev)


def step():
    sync.zero_grad()
    vloss.mpjpe(m(x), tgt).backward()
    sync.sync()


def timed(n=30):
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(3):
    step()
mode["v"] = "record"
step()
torch.cuda.synchronize()
mode["v"] = "real"
print("recorded %d BatchNorm-backward reductions per step" % len(replay), flush=True)

res = {"real": [], "removed": []}
for rep in range(4):
    for v in ("real", "removed"):
        mode["v"] = v
        calls["n"] = 0
        res[v].append(timed())
print("step with the reductions : %s  (min %.3f ms)" % (" ".join("%.3f" % t for t in res["real"]), min(res["real"])))
print("step without them        : %s  (min %.3f ms)   [%d replaced launches per step]" % (
    " ".join("%.3f" % t for t in res["removed"]), min(res["removed"]), calls["n"] // 34))
print("ceiling of any fusion    : %.3f ms per step (+ <= 0.02 ms for the 8 copy launches that stand in)" % (
    min(res["real"]) - min(res["removed"])))
