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
from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world, local_rank).  A single process without those variables is world 1 (no group)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


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
    is a single large all-reduce (xGMI rings are per-link bound: few large messages, not many small ones)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], world: Optional[int] = None, group=None,
                 direct_module: Optional[torch.nn.Module] = None):
        """direct_module: a videopose3d_amd model whose backward should WRITE its conv-weight gradients straight
        into the flat buffer (no autograd accumulation pass).  Requires zero_grad() before every backward; gradient
        accumulation over several backward passes is then not supported for those tensors."""
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameters"
        dev, dt = self.params[0].device, self.params[0].dtype
        assert all(p.device == dev and p.dtype == dt for p in self.params)
        self.offsets, self.numel = flat_layout(self.params)      # numel includes the alignment padding (zeros)
        self.flat = torch.zeros(self.numel, dtype=dt, device=dev)
        self.group = group
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self._attach()
        self._views = {id(p): p.grad for p in self.params}
        if direct_module is not None:
            direct_module.__dict__["_vp3d_grad_sink"] = self

    def view_for(self, p: torch.nn.Parameter) -> Optional[torch.Tensor]:
        """The flat-buffer view that is (and stays) p.grad, or None when p is not managed / was detached."""
        v = self._views.get(id(p))
        return v if (v is not None and p.grad is not None and p.grad.data_ptr() == v.data_ptr()) else None

    def _attach(self):
        for p, off in zip(self.params, self.offsets):
            p.grad = self.flat[off:off + p.numel()].view_as(p)

    def zero_grad(self):
        """Use instead of optimizer.zero_grad(set_to_none=True), which would detach the views."""
        self.flat.zero_()
        if any(p.grad is None for p in self.params):
            self._attach()
            self._views = {id(p): p.grad for p in self.params}

    def broadcast_parameters(self, buffers: Iterable[torch.Tensor] = ()):
        """Make every replica start from rank 0's weights (and BN buffers)."""
        if self.world > 1:
            for t in list(self.params) + list(buffers):
                dist.broadcast(t.data if isinstance(t, torch.nn.Parameter) else t, src=0, group=self.group)

    def sync(self, local_count: Optional[int] = None, global_count: Optional[int] = None):
        """Sum-all-reduce the flat gradients and turn the sum into the global-batch mean.

        With equal per-rank batch sizes the result is sum/world.  For the short last batch pass the local and
        global sample counts: each rank's mean-loss gradient is re-weighted by local_count/global_count."""
        if self.world == 1:
            return
        if local_count is not None and global_count is not None:
            self.flat.mul_(float(local_count) / float(global_count))
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(self.world)
