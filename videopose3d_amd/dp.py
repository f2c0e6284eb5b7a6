"""Data parallelism for the training path: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" on CPU for tests).

The reference is single-process (no distributed code at all).  The shard point is the
batch dimension of ``ChunkedGenerator`` (generators.py:99-160): every rank builds the identical generator (same
``random_seed=1234`` => identical permutation, no communication) and takes its contiguous slice of each global
batch; samples are independent in forward/backward except through (i) BatchNorm batch statistics, kept
per-replica (each rank sees the reference's own 1024-sample statistics regime) and (ii) the parameter-gradient
sum, which is the ONE exchange step: a single sum all-reduce of a flat fp32 gradient buffer (16,952,371 floats =
67.8 MB for arc 3,3,3,3,3 / C=1024), then a divide by the world size (mpjpe is a mean, loss.py:17).
"""
from __future__ import annotations

import os
import sys
from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


# RCCL channel budget of the overlapped bucket all-reduce.  Every RCCL channel is one workgroup that stays RESIDENT on a CU for
# the whole collective, and the 256 x 256 / 224 x 256 GEMM workgroups of the backward need a CU's entire register file: a CU
# that hosts a channel is lost to them, and kernels laid out for exactly 256 CUs (the expand layer's persistent kernels, whole
# rounds of tiles) pay a whole extra round.  Measured on one MI355X with N resident stand-in workgroups on a third queue
# (tools/cu_hog_ab.py, profiles/r04_cu_hog_interference.txt): 8 -> +14.5 % step time while they are resident, 16 -> +36 %,
# 32 -> +133 %.  Eight channels move a 17 MB bucket in ~0.15 ms (ring over xGMI, ~30 GB/s per channel workgroup), i.e. the four
# buckets are resident for ~0.6 ms of the 2.6 ms backward: ~+0.09 ms per step, against ~0.45 ms for an exchange that is not
# overlapped at all and ~+0.3 ms for RCCL's default of 32+ channels.  The user's own NCCL_* settings win (setdefault).
RCCL_ENV_DEFAULTS = {"NCCL_MAX_NCHANNELS": "8", "NCCL_MIN_NCHANNELS": "4"}


def pin_rank_to_cores(local_rank: int, local_world: int):
    """Give every rank of a node its own slice of the cores this process may run on (the step's host side is one Python
    thread enqueueing ~2 ms of launches per 4.4 ms step: N unpinned ranks migrate over each other's cores and caches).
    Returns the sorted core list the rank is pinned to, or None when there is nothing to split (fewer cores than ranks, no
    sched_setaffinity)."""
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    if local_world <= 1 or len(cores) < local_world:
        return None
    per = len(cores) // local_world
    mine = cores[local_rank * per:(local_rank + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    torch.set_num_threads(max(1, min(per, 4)))
    return mine


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world, local_rank).  A single process without those variables is world 1 (no group).  For world > 1 the
    RCCL channel budget (RCCL_ENV_DEFAULTS) is put into the environment before the communicator is created -- as DEFAULTS
    (values already in the environment win), announced once on rank 0's stderr because the setting is process-wide (it also
    caps sync-BN's all_gather, broadcast_parameters and the user's own collectives), and skipped altogether with
    VP3D_RCCL_CHANNELS=0.  The numbers are a guess with a measured lower bound: 8 resident stand-in workgroups cost the backward
    GEMMs +14.5 % on one GPU (profiles/r04_cu_hog_interference.txt); real RCCL kernels over xGMI have not been measurable here."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("VP3D_RCCL_CHANNELS", "1") != "0":
            for k, v in RCCL_ENV_DEFAULTS.items():
                os.environ.setdefault(k, v)
            if rank == 0:
                print("[videopose3d_amd.dp] RCCL channel budget for this process: %s (VP3D_RCCL_CHANNELS=0 leaves RCCL's own defaults)"
                      % ", ".join("%s=%s" % (k, os.environ[k]) for k in RCCL_ENV_DEFAULTS), file=sys.stderr, flush=True)
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


MIN_SHARD = 2            # training-mode BatchNorm needs more than one row per channel (T_out = 1: rows == samples)


def shardable(n_items: int, world: int) -> bool:
    """True when a global batch of n_items gives EVERY rank at least MIN_SHARD samples.  The short last batch of an
    epoch (generators.py:57,104) may not: sharded generators drop such a batch on all ranks alike (a rank with 0 or 1
    samples cannot run the training step, and a rank that skips backward must not skip the collectives)."""
    return n_items >= MIN_SHARD * world


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of a global batch of n_items for `rank` (balanced; the short last batch of an
    epoch, generators.py:57,104, gives some ranks one item fewer)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(arrays: Sequence, rank: int, world: int) -> List:
    """Slice every array of a ChunkedGenerator batch (cam, batch_3d, batch_2d; any may be None) along dim 0."""
    n = next(a.shape[0] for a in arrays if a is not None)
    lo, hi = shard_bounds(n, rank, world)
    return [None if a is None else a[lo:hi] for a in arrays]


FLAT_ALIGN = 64          # floats (256 B): every tensor's slot in a flat buffer starts on a 256-byte boundary, so the
#                          views satisfy the 16-byte alignment of the float4 / LDS-DMA kernel paths


def flat_layout(params: Sequence[torch.Tensor], align: int = FLAT_ALIGN) -> Tuple[List[int], int]:
    """Offsets (in elements) of each tensor inside a flat buffer and the padded total length."""
    offs, off = [], 0
    for p in params:
        offs.append(off)
        off += (p.numel() + align - 1) // align * align
    return offs, off


class FlatGradSync:
    """Keeps every parameter's ``.grad`` as a view into ONE contiguous fp32 buffer so that the gradient exchange
    is a few large all-reduces (xGMI rings are per-link bound: few large messages, not many small ones).

    With ``direct_module`` (a videopose3d_amd model) the flat buffer is laid out in the order in which the hand-written
    backward FINISHES the gradients (shrink, last block, ..., first block, expand) and cut into buckets of
    >= ``bucket_bytes``; the engine reports every finished group (``group_done``) and each complete bucket is
    all-reduced asynchronously (RCCL runs it on its own stream) while backward keeps computing the earlier layers:
    for arc 3,3,3,3,3 the last three blocks (50 MB of the 67.8 MB) are exchanged underneath the two M = 27,648
    layers that hold 60 % of the backward FLOPs; only the expand layer's 0.4 MB is exposed."""

    def __init__(self, params: Iterable[torch.nn.Parameter], world: Optional[int] = None, group=None,
                 direct_module: Optional[torch.nn.Module] = None, bucket_bytes: int = 16 << 20,
                 always_reduce: bool = False):
        """direct_module: a videopose3d_amd model whose backward should WRITE its gradients straight into the flat
        buffer (no autograd accumulation pass) and report finished groups for the overlapped exchange.  Requires
        zero_grad() before every backward; gradient accumulation over several backward passes is then not supported.
        always_reduce: issue the collectives even when world == 1 (tests of the RCCL plumbing on one GPU)."""
        plist = [p for p in params if p.requires_grad]
        assert plist, "no trainable parameters"
        self._group_sizes: List[int] = []
        if direct_module is not None and hasattr(direct_module, "backward_param_groups"):
            groups = [[p for p in g if p.requires_grad] for g in direct_module.backward_param_groups()]
            ordered = [p for g in groups for p in g]
            if {id(p) for p in ordered} == {id(p) for p in plist} and len(ordered) == len(plist):
                plist = ordered
                self._group_sizes = [len(g) for g in groups]
        self.params = plist
        dev, dt = self.params[0].device, self.params[0].dtype
        assert all(p.device == dev and p.dtype == dt for p in self.params)
        self.offsets, self.numel = flat_layout(self.params)      # numel includes the alignment padding (zeros)
        self.flat = torch.zeros(self.numel, dtype=dt, device=dev)
        self.group = group
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self._reduce = self.world > 1 or always_reduce
        self._attach()
        self._views = {id(p): p.grad for p in self.params}
        # buckets: [first_group, last_group] -> flat range; merge consecutive backward groups up to bucket_bytes
        self.buckets: List[Tuple[int, int, int]] = []            # (last_group, lo, hi) in flat elements
        if self._group_sizes:
            esz = self.flat.element_size()
            starts, k = [], 0
            for n in self._group_sizes:
                starts.append(self.offsets[k])
                k += n
            starts.append(self.numel)
            lo_g = 0
            for g in range(len(self._group_sizes)):
                if (starts[g + 1] - starts[lo_g]) * esz >= bucket_bytes or g == len(self._group_sizes) - 1:
                    self.buckets.append((g, starts[lo_g], starts[g + 1]))
                    lo_g = g + 1
        self._seen = set()           # backward groups reported since zero_grad() (a repeat = a second backward pass)
        self._done = 0               # groups finished in the current backward
        self._launched = 0           # buckets already handed to the collective
        self._handles = []
        self._weight = None          # local_count / global_count of the current step (short last batch)
        if direct_module is not None:
            direct_module.__dict__["_vp3d_grad_sink"] = self

    def view_for(self, p: torch.nn.Parameter) -> Optional[torch.Tensor]:
        """The flat-buffer view that is (and stays) p.grad, or None when p is not managed / was detached."""
        v = self._views.get(id(p))
        return v if (v is not None and p.grad is not None and p.grad.data_ptr() == v.data_ptr()) else None

    def _attach(self):
        for p, off in zip(self.params, self.offsets):
            p.grad = self.flat[off:off + p.numel()].view_as(p)

    def zero_grad(self, local_count: Optional[int] = None, global_count: Optional[int] = None):
        """Use instead of optimizer.zero_grad(set_to_none=True), which would detach the views.  Starts a new step:
        pass the local / global sample counts of a short last batch here (each rank's mean-loss gradient is then
        re-weighted by local_count/global_count before it is summed)."""
        assert not self._handles, "zero_grad() while a gradient exchange is in flight: call sync() first"
        self.flat.zero_()
        self._seen.clear()
        if any(p.grad is None for p in self.params):
            self._attach()
            self._views = {id(p): p.grad for p in self.params}
        self._done = self._launched = 0
        self._weight = (float(local_count) / float(global_count)) if (local_count is not None and
                                                                       global_count is not None) else None

    def broadcast_parameters(self, buffers: Iterable[torch.Tensor] = ()):
        """Make every replica start from rank 0's weights (and BN buffers)."""
        if self.world > 1:
            for t in list(self.params) + list(buffers):
                dist.broadcast(t.data if isinstance(t, torch.nn.Parameter) else t, src=0, group=self.group)

    # ---- overlapped exchange ---------------------------------------------------------------------------
    def _launch(self, lo: int, hi: int):
        seg = self.flat[lo:hi]
        if self._weight is not None:
            seg.mul_(self._weight)
        self._handles.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def group_done(self, k: int):
        """Called by the engine's backward when every gradient of backward group k has been written."""
        if k in self._seen:
            # the direct sink OVERWRITES the flat views: a second backward pass before zero_grad() (two model calls in
            # one step, gradient accumulation) would silently drop the first pass's gradients
            from ._lib import Vp3dError
            raise Vp3dError("FlatGradSync(direct_module=...): a second backward pass reached the gradient sink before "
                            "zero_grad(); the direct sink overwrites gradients (no accumulation). Call zero_grad() before "
                            "every backward, or build the FlatGradSync without direct_module")
        self._seen.add(k)
        if not self._reduce or not self.buckets:
            return
        self._done = max(self._done, k + 1)
        while self._launched < len(self.buckets) and self.buckets[self._launched][0] < self._done:
            _, lo, hi = self.buckets[self._launched]
            self._launch(lo, hi)
            self._launched += 1

    def sync(self, local_count: Optional[int] = None, global_count: Optional[int] = None):
        """Finish the sum-all-reduce of the flat gradients and turn the sum into the global-batch mean.

        With equal per-rank batch sizes the result is sum/world.  For the short last batch pass the local and
        global sample counts (here, or to zero_grad() when the overlapped exchange is used): each rank's
        mean-loss gradient is re-weighted by local_count/global_count."""
        if not self._reduce:
            return
        if local_count is not None and global_count is not None:
            w = float(local_count) / float(global_count)
            assert self._launched == 0 or self._weight == w, \
                "buckets were already exchanged during backward: pass the sample counts to zero_grad() instead"
            self._weight = w
        # Every rank must issue the SAME sequence of collectives whatever its backward reported (a rank that skipped
        # backward, e.g. on an empty shard, reports nothing): with buckets the exchange is always bucket by bucket.
        if not self.buckets:
            self._launch(0, self.numel)                      # no backward-order layout: one whole-buffer all-reduce
        else:
            while self._launched < len(self.buckets):        # buckets the backward did not (yet) hand over
                _, lo, hi = self.buckets[self._launched]
                self._launch(lo, hi)
                self._launched += 1
        for h in self._handles:
            h.wait()
        self._handles = []
        if self._weight is None:
            self.flat.div_(self.world)
        self._done = self._launched = 0
        self._weight = None


class SyncBatchNorm:
    """Optional synchronised BatchNorm: batch statistics over the GLOBAL batch instead of per replica, which
    makes N ranks x B/N samples reproduce a single-device step on B samples.  Per BatchNorm layer: one all-gather of
    [2, C] (mean, M2 -- merged with Chan's formula in fp64, no E[x^2] - E[x]^2 cancellation) in forward and one all-reduce
    of [2, C] (sum g, sum g*xhat) in backward: 18 + 18 small collectives per step for arc 3,3,3,3,3 (latency-bound; use
    it to prove equivalence or for very small per-rank batches, not for speed).  Running statistics are then identical on
    all ranks.  Enable with ``SyncBatchNorm(model)`` after ``init_from_env``; ``model.__dict__["_vp3d_sync_bn"]``
    is what the engines look for."""

    def __init__(self, model: torch.nn.Module, group=None):
        assert dist.is_initialized(), "SyncBatchNorm needs an initialised process group"
        self.group = group
        self.world = dist.get_world_size(group)
        self.frac = 1.0 / self.world          # local rows / global rows of the current step (begin_step refines it)
        model.__dict__["_vp3d_sync_bn"] = self

    def begin_step(self, batch_local: int, device) -> None:
        t = torch.tensor([float(batch_local)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        self.frac = float(batch_local) / float(t.item())

    def rows_total(self, rows_local: int) -> int:
        return int(round(rows_local / self.frac))

    def merge_stats(self, mean: torch.Tensor, var: torch.Tensor, n_local: int):
        """(mean, biased var) over n_local rows per rank -> global (mean, biased var, n_total), fp64 Chan merge."""
        c = mean.numel()
        mine = torch.empty(2 * c + 1, dtype=torch.float64, device=mean.device)
        mine[:c] = mean.double()
        mine[c:2 * c] = var.double() * n_local
        mine[2 * c] = float(n_local)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        allp = torch.stack(parts)                                   # [world, 2C+1]
        n = allp[:, 2 * c]
        n_tot = n.sum()
        mean_g = (allp[:, :c] * n[:, None]).sum(0) / n_tot
        m2_g = (allp[:, c:2 * c] + n[:, None] * (allp[:, :c] - mean_g[None, :]) ** 2).sum(0)
        return mean_g, m2_g / n_tot, n_tot

    def sum_(self, t: torch.Tensor) -> torch.Tensor:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t
