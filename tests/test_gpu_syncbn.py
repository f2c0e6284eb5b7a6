"""dp.SyncBatchNorm (optional): two ranks x 8 samples with synchronised BatchNorm statistics must reproduce a
single-process step on the 16 samples -- outputs, every (averaged) gradient, running statistics.  The two ranks share the
one GPU of the test box and talk through gloo (RCCL refuses two ranks on one device); the collectives are the same calls."""
import os
import socket
import subprocess
import sys

import pytest
import torch

import videopose3d_amd as V
from videopose3d_amd import loss as vloss

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("math", ["f16x3", "f32"])
def test_two_ranks_with_sync_bn_equal_one_big_batch(math, tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "rank0.pt")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   VP3D_S16_MIN_GFLOP="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_syncbn_worker.py"), out, math], env=env))
    for p in procs:
        assert p.wait(timeout=500) == 0
    got = torch.load(out)
    # single process, whole batch
    from videopose3d_amd import engine
    keep = dict(engine.S16_MIN_FORWARD_FLOPS)
    engine.S16_MIN_FORWARD_FLOPS.update({True: 0.0, False: 0.0})
    try:
        dev = "cuda:0"
        torch.manual_seed(7)
        m = V.TemporalModelOptimized1f(17, 2, 17, [3, 3, 3], dropout=0.0, channels=128).to(dev).train()
        m.math = math
        gen = torch.Generator().manual_seed(11)
        x = (torch.randn(16, 27, 17, 2, generator=gen) * 0.5).clamp(-1, 1)
        tgt = torch.randn(16, 1, 17, 3, generator=gen) * 0.3
        y = m(x.to(dev))
        vloss.mpjpe(y, tgt.to(dev)).backward()
    finally:
        engine.S16_MIN_FORWARD_FLOPS.update(keep)
    assert float((got["y"] - y[:8].detach().cpu()).abs().max()) < 2e-5
    for k, p in m.named_parameters():
        ref = p.grad.cpu()
        err = float((got["grads"][k] - ref).abs().max() / (ref.abs().max() + 1e-30))
        assert err < 2e-4, (k, err)
    for k, b in m.named_buffers():
        if b.dtype.is_floating_point:
            assert torch.allclose(got["buffers"][k], b.cpu(), rtol=1e-5, atol=1e-6), k
        else:
            assert int(got["buffers"][k]) == int(b)
