#!/usr/bin/env python3
"""What do N resident workgroups on another queue -- RCCL's channel kernels during an overlapped bucket all-reduce -- cost the
training step?  One GPU, no collective: tools/ubench/cu_hog.hip keeps N workgroups (256 threads, 16 KiB LDS, a 64-KiB copy loop
each) resident on a third stream while the benchmark step runs.  A 256 x 256 / 224 x 256 GEMM workgroup takes a CU's whole
register file (8 waves x 254 VGPRs), so a CU that hosts a hog workgroup is LOST to those GEMMs until the hog leaves
(profiles/r03_coexist_ubench.txt: co-residency is a matter of registers) -- the step then sees 256 - N CUs.

    python tools/cu_hog_ab.py [hog counts ...]        default: 0 8 16 32 64
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402
from videopose3d_amd import dp, loss as vloss  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(HERE, "ubench", "libcuhog.so"))
lib.cu_hog_launch.restype = C.c_int
lib.cu_hog_launch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_int]

counts = [int(v) for v in sys.argv[1:]] or [0, 8, 16, 32, 64]
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


def fwd():
    m(x)


hog_stream = torch.cuda.Stream(device=dev)
stop = torch.zeros(1, dtype=torch.int32).pin_memory()
SLICE = 1 << 20                                                   # 4 MiB of buffer per hog workgroup
buf = torch.zeros(max(counts + [1]) * SLICE, dtype=torch.float32, device=dev)
where = torch.zeros(max(counts + [1]) * 2, dtype=torch.int32, device=dev)


def timed(fn, n, hogs):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    stop[0] = 0
    if hogs:
        rc = lib.cu_hog_launch(hog_stream.cuda_stream, hogs, stop.data_ptr(), buf.data_ptr(), SLICE, where.data_ptr(), 2000.0, 1)
        assert rc == 0, rc
        time.sleep(0.01)                                           # the hogs are resident before the first step is enqueued
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.current_stream().synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    stop[0] = 1
    torch.cuda.synchronize()
    return dt


print("hog workgroups resident on a third stream (each: 256 threads, 16 KiB LDS, 64-KiB copy loop) vs the cfg3 step, B = 1024")
base = {}
for name, fn in (("whole step", step), ("forward only", fwd)):
    res = {h: [] for h in counts}
    for rep in range(4):
        for h in counts:
            res[h].append(timed(fn, 25, h))
    for h in counts:
        med = sorted(res[h])[len(res[h]) // 2]
        base.setdefault(name, med if h == 0 else None)
        line = "%-12s hogs %3d: %s  median %.3f ms" % (name, h, " ".join("%.3f" % t for t in res[h]), med)
        if h and base.get(name):
            line += "  (%+.1f %% vs none; CUs lost if proportional: %.1f %%)" % ((med / base[name] - 1) * 100, h / 256 * 100)
        print(line, flush=True)
w = where.cpu().view(-1, 2)[:max(counts)]
cus = set((int(a), int(b) & 0xf00 | (int(b) >> 13 & 0x7) << 4 | (int(b) >> 8 & 0xf)) for a, b in w.tolist())
print("placement of the last %d-hog launch: %d distinct (XCC, HW_ID CU/SE fields) pairs; XCC histogram %s" % (
    max(counts), len(cus), [sum(1 for a, _ in w.tolist() if a == k) for k in range(8)]))
