#!/usr/bin/env python3
"""Does a hipGraph replay of the cfg3 step keep the second-stream overlap of the eager step?

    python tools/graph_branches.py [dot_path]

Captures graph.GraphedTrainStep with torch's debug mode on, writes hipGraphDebugDotPrint's .dot (the fork / join edges of
the two-stream backward are visible there), counts nodes with more than one successor, and times eager vs replay
(interleaved).  Run it once per HIP-runtime setting -- the knobs are read at HIP initialisation:
    DEBUG_HIP_FORCE_GRAPH_QUEUES=N      streams the graph executor may use for parallel branches
    DEBUG_CLR_GRAPH_PACKET_CAPTURE=0/1  pre-recorded AQL packets for kernel nodes
"""
import os
import re
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import videopose3d_amd as V  # noqa: E402
from videopose3d_amd import dp, loss as vloss  # noqa: E402
from videopose3d_amd import graph as G  # noqa: E402

dot = sys.argv[1] if len(sys.argv) > 1 else None
dev = "cuda:0"
torch.manual_seed(0)
x = (torch.randn(1024, 243, 17, 2, device=dev) * 0.5).clamp(-1, 1)
tgt = torch.randn(1024, 1, 17, 3, device=dev) * 0.3
m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3, 3, 3], dropout=0.25, channels=1024).to(dev).train()
sync = dp.FlatGradSync(m.parameters(), world=1, direct_module=m)


def eager():
    sync.zero_grad()
    vloss.mpjpe(m(x), tgt).backward()
    sync.sync()


def timed(fn, n=25):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


if dot:
    G.DEBUG_DOT_PATH = dot
step = G.GraphedTrainStep(m, sync)
step(x, tgt)
torch.cuda.synchronize()
if dot and os.path.exists(dot):
    txt = open(dot).read()
    edges = re.findall(r'"?([\w.]+)"?\s*->\s*"?([\w.]+)"?', txt)
    succ, pred = {}, {}
    for a, b in edges:
        succ.setdefault(a, set()).add(b)
        pred.setdefault(b, set()).add(a)
    print("dot: %d edges, %d nodes with >1 successor (forks), %d with >1 predecessor (joins)"
          % (len(edges), sum(len(v) > 1 for v in succ.values()), sum(len(v) > 1 for v in pred.values())), flush=True)
env = {k: os.environ.get(k) for k in ("DEBUG_HIP_FORCE_GRAPH_QUEUES", "DEBUG_CLR_GRAPH_PACKET_CAPTURE", "VP3D_OVERLAP")}
for rep in range(3):
    print("%s  eager %.3f ms   graph replay %.3f ms" % (env, timed(eager), timed(lambda: step(x, tgt))), flush=True)
