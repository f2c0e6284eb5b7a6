"""Worker of tests/test_gpu_syncbn.py: rank r of a world-2 gloo group on ONE GPU runs a training step on its half of the
batch with dp.SyncBatchNorm + dp.FlatGradSync and rank 0 saves what the parent compares with a single-process step."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import videopose3d_amd as V  # noqa: E402
from videopose3d_amd import dp  # noqa: E402
from videopose3d_amd import loss as vloss  # noqa: E402


def main():
    out_path, math = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = "cuda:0"
    torch.manual_seed(7)
    m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3], dropout=0.0, channels=128).to(dev).train()
    m.math = math
    gen = torch.Generator().manual_seed(11)
    x = (torch.randn(16, 27, 17, 2, generator=gen) * 0.5).clamp(-1, 1)
    tgt = torch.randn(16, 1, 17, 3, generator=gen) * 0.3
    lo, hi = dp.shard_bounds(16, rank, world)
    dp.SyncBatchNorm(m)
    sync = dp.FlatGradSync(m.parameters(), direct_module=m)
    sync.broadcast_parameters(m.buffers())
    sync.zero_grad()
    y = m(x[lo:hi].to(dev))
    vloss.mpjpe(y, tgt[lo:hi].to(dev)).backward()
    sync.sync()
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"y": y.detach().cpu(), "grads": {k: p.grad.cpu() for k, p in m.named_parameters()},
                    "buffers": {k: b.cpu() for k, b in m.named_buffers()}}, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
