#!/usr/bin/env python3
"""Evaluation throughput over a set of videos: one sequence (+ mirrored copy) per call, as run.py does, vs
generators.predict_sequences (length-grouped batches)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402
from videopose3d_amd import generators as G  # noqa: E402

dev = "cuda:0"
rng = np.random.RandomState(0)
lens = [int(x) for x in rng.randint(900, 3500, size=120)]                 # Human3.6M-like test videos
p2 = [rng.standard_normal((n, 17, 2)).astype(np.float32) * 0.5 for n in lens]
kl, kr = [1, 3, 5, 7, 9, 11, 13, 15], [2, 4, 6, 8, 10, 12, 14, 16]
jl, jr = [4, 5, 6, 11, 12, 13], [1, 2, 3, 14, 15, 16]
for math in ("f32", "f16x3"):
    m = V.TemporalModel(17, 2, 17, [3, 3, 3, 3, 3], channels=1024).to(dev).eval()
    m.math = math
    gen = G.UnchunkedGenerator(None, None, p2, pad=121, augment=True, kps_left=kl, kps_right=kr, joints_left=jl,
                               joints_right=jr, device=dev)

    def one_by_one():
        with torch.no_grad():
            return [G.tta_average(m(b2), jl, jr)[0] for _, _, b2 in gen.next_epoch()]
    for fn, tag in ((one_by_one, "one video per call"), (lambda: G.predict_sequences(m, gen, 65536), "length-grouped batches")):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print("math=%-6s %-24s %7.1f ms for %d videos / %d frames  -> %.2f M frames/s" % (
            math, tag, dt * 1e3, len(lens), sum(lens), sum(lens) / dt / 1e6), flush=True)
