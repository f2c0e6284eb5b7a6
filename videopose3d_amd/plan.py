"""Layer plan of the temporal model: pure shape / index arithmetic (no tensors, no GPU).

Restates the hyper-parameter logic of the reference constructors (common/model.py:85-124 for the dilated
``TemporalModel`` incl. ``dense``, :151-185 for the strided ``TemporalModelOptimized1f``) and the residual
slices of ``_forward_blocks`` (:132 crop, :191 strided pick) as data that the engine turns into GEMM row maps.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List


@dataclass(frozen=True)
class ConvSpec:
    """One temporal convolution: y[b,t] = sum_k x[b, t*stride + k*dil] @ W_k."""
    c_in: int
    c_out: int
    taps: int
    dil: int = 1
    stride: int = 1

    def t_out(self, t_in: int) -> int:
        return (t_in - self.dil * (self.taps - 1) - 1) // self.stride + 1


@dataclass(frozen=True)
class ResSpec:
    """Residual rows of a block: output row t adds input row t*step + start (model.py:132 / :191)."""
    start: int
    step: int


@dataclass(frozen=True)
class StackPlan:
    kind: str                    # "dilated" | "strided"
    filter_widths: tuple
    causal: bool
    dense: bool
    pad: tuple
    causal_shift: tuple
    convs: tuple                 # ConvSpec for expand, then (wide, pointwise) per block
    res: tuple                   # ResSpec per block
    shrink: ConvSpec

    @property
    def n_blocks(self) -> int:
        return len(self.res)

    def receptive_field(self) -> int:
        return 1 + 2 * sum(self.pad)                                   # model.py:41-48

    def total_causal_shift(self) -> int:
        frames = self.causal_shift[0]                                  # model.py:50-61, as written
        next_dil = self.filter_widths[0]
        for i in range(1, len(self.filter_widths)):
            frames += self.causal_shift[i] * next_dil
            next_dil *= self.filter_widths[i]
        return frames

    def forward_flops(self, batch: int, t_in: int) -> float:
        """Conv FLOPs (2 per MAC) of one forward pass over [batch, t_in] frames (shrink included)."""
        t, total = t_in, 0.0
        for spec in self.convs:
            t = spec.t_out(t) if spec is self.convs[0] or spec.taps > 1 else t
            total += 2.0 * batch * max(t, 0) * spec.c_out * spec.c_in * spec.taps
        return total + 2.0 * batch * max(t, 0) * self.shrink.c_out * self.shrink.c_in

    def lengths(self, t_in: int) -> List[int]:
        """Time length after expand and after every block; raises if the input is shorter than needed."""
        t = self.convs[0].t_out(t_in)
        out = [t]
        for i in range(self.n_blocks):
            t = self.convs[1 + 2 * i].t_out(t)
            out.append(t)
        if min(out) < 1:
            raise ValueError("input sequence of %d frames is shorter than the receptive field (%d)"
                             % (t_in, self.receptive_field()))
        return out


def make_plan(kind: str, in_channels: int, channels: int, out_channels: int, filter_widths, causal=False,
              dense=False) -> StackPlan:
    fw = tuple(int(f) for f in filter_widths)
    for f in fw:
        assert f % 2 != 0, "Only odd filter widths are supported"      # model.py:20-21
    assert kind in ("dilated", "strided")
    strided = kind == "strided"
    assert not (strided and dense)
    pad = [fw[0] // 2]
    shift = [fw[0] // 2 if causal else 0]
    convs = [ConvSpec(in_channels, channels, fw[0], 1, fw[0] if strided else 1)]
    res = []
    next_dil = fw[0]
    for i in range(1, len(fw)):
        pad.append((fw[i] - 1) * next_dil // 2)
        if strided:
            shift.append(fw[i] // 2 if causal else 0)
            convs.append(ConvSpec(channels, channels, fw[i], 1, fw[i]))
            res.append(ResSpec(start=shift[-1] + fw[i] // 2, step=fw[i]))
        else:
            shift.append((fw[i] // 2 * next_dil) if causal else 0)
            if dense:
                convs.append(ConvSpec(channels, channels, 2 * pad[-1] + 1, 1, 1))
            else:
                convs.append(ConvSpec(channels, channels, fw[i], next_dil, 1))
            res.append(ResSpec(start=pad[-1] + shift[-1], step=1))
        convs.append(ConvSpec(channels, channels, 1, 1, 1))
        next_dil *= fw[i]
    return StackPlan(kind, fw, bool(causal), bool(dense), tuple(pad), tuple(shift), tuple(convs), tuple(res),
                     ConvSpec(channels, out_channels, 1, 1, 1))
