import sys, torch, time
sys.path.insert(0, ".")
import videopose3d_amd as V
from videopose3d_amd import loss as vloss
dev="cuda:0"
torch.manual_seed(0)
m = V.TemporalModelOptimized1f(17, 2, 17, [3,3,3,3,3], dropout=0.25, channels=1024).to(dev).train()
for b in (2048, 8192):
    x = (torch.randn(b, 243, 17, 2, device=dev) * 0.5).clamp(-1, 1)
    tgt = torch.randn(b, 1, 17, 3, device=dev) * 0.3
    m.zero_grad(set_to_none=True)
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        torch.cuda.synchronize(); t0=time.perf_counter()
        l = vloss.mpjpe(m(x), tgt); l.backward()
        torch.cuda.synchronize(); dt=time.perf_counter()-t0
    g = m.expand_conv.weight.grad
    print("B=%d loss %.5f  |dW0| %.4e finite %s  %.2f ms -> %.0f frames/s" % (b, float(l), float(g.abs().max()), bool(torch.isfinite(g).all()), dt*1e3, b/dt))
    del x, tgt
