"""The reference's CPU execution path, restated.  TEST / BASELINE INFRASTRUCTURE ONLY.

``temporal_oracle.py`` is the parity checker (independent numpy arithmetic).  This file exists for ONE purpose:
``bench.py``'s ``cpu_baseline`` leg needs "the reference's CPU path timed on the same host", and
/root/reference does not exist on the GPU box.  What the reference executes on a CPU is ATen/oneDNN through
``torch.nn.Conv1d / BatchNorm1d / ReLU / Dropout`` + autograd, so this restatement drives exactly those library
kernels through ``torch.nn.functional`` with the wiring of

  reference common/model.py:63-77    forward(): [B,T,J,F] -> view [B,T,J*F] -> permute(0,2,1) -> blocks -> permute back
  reference common/model.py:126-138  TemporalModel._forward_blocks (dilated convs, centre-crop residual)
  reference common/model.py:187-197  TemporalModelOptimized1f._forward_blocks (strided convs, strided residual)
  reference common/loss.py:11-17     mpjpe

operating directly on a reference-format ``state_dict`` (no nn.Module, no import of reference code).
It is pinned against the reference-generated fixtures in tests/golden by tests/test_oracle_golden.py.
Only tests/ and bench.py's cpu_baseline leg may import it; the product (videopose3d_amd) never does.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _bn(x, sd, prefix, training, momentum):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                        sd[prefix + ".bias"], training, momentum, 1e-5)


def forward(sd, x, filter_widths, *, kind="dilated", causal=False, dense=False, training=False, dropout=0.0,
            momentum=0.1):
    """x [B,T,J,F] float32 CPU tensor -> [B,T_out,J_out,3].  ``sd``: reference state_dict (tensors; the BN running
    buffers are updated in place when ``training``)."""
    fw = list(filter_widths)
    strided = kind == "strided"
    b, t = x.shape[0], x.shape[1]
    h = x.reshape(b, t, -1).permute(0, 2, 1)                                   # model.py:68-70

    def act(v, prefix):                                                          # drop(relu(bn(.)))
        v = F.relu(_bn(v, sd, prefix, training, momentum))
        return F.dropout(v, dropout, training) if dropout > 0 else v

    h = act(F.conv1d(h, sd["expand_conv.weight"], stride=fw[0] if strided else 1), "expand_bn")   # :127 / :188
    next_dil = fw[0]
    for i in range(1, len(fw)):
        w0, w1 = sd["layers_conv.%d.weight" % (2 * (i - 1))], sd["layers_conv.%d.weight" % (2 * (i - 1) + 1)]
        if strided:
            shift = fw[i] // 2 if causal else 0
            res = h[:, :, shift + fw[i] // 2::fw[i]]                           # :191
            u = F.conv1d(h, w0, stride=fw[i])                                  # :193
        else:
            pad = (fw[i] - 1) * next_dil // 2
            shift = (fw[i] // 2 * next_dil) if causal else 0
            res = h[:, :, pad + shift: h.shape[2] - pad + shift]               # :132
            u = F.conv1d(h, w0, dilation=1 if dense else next_dil)             # :134 (dense: 2*pad+1 taps, dil 1)
        u = act(u, "layers_bn.%d" % (2 * (i - 1)))
        h = res + act(F.conv1d(u, w1), "layers_bn.%d" % (2 * (i - 1) + 1))     # :135 / :194
        next_dil *= fw[i]
    out = F.conv1d(h, sd["shrink.weight"], sd["shrink.bias"])                  # :137 / :196
    out = out.permute(0, 2, 1)                                                 # model.py:73-75
    return out.reshape(b, -1, out.shape[2] // 3, 3)


def mpjpe(pred, target):                                                       # loss.py:11-17
    return torch.mean(torch.norm(pred - target, dim=len(target.shape) - 1))


PARAM_SUFFIXES = (".weight", ".bias")


def train_step(sd, x, target, filter_widths, **kw):
    """One forward + backward of the mpjpe loss; returns (loss, out, {name: grad})."""
    names = [k for k in sd if k.endswith(PARAM_SUFFIXES) and sd[k].is_floating_point()]
    leaf = dict(sd)
    for k in names:
        leaf[k] = sd[k].detach().clone().requires_grad_(True)
    out = forward(leaf, x, filter_widths, training=True, **kw)
    loss = mpjpe(out, target)
    grads = torch.autograd.grad(loss, [leaf[k] for k in names])
    return loss.detach(), out.detach(), dict(zip(names, grads))
