"""videopose3d_amd -- MI355X (gfx950) native implementation of the VideoPose3D temporal-model hot path.

Public surface = the reference's (common/model.py): ``TemporalModel``, ``TemporalModelOptimized1f``,
``TemporalModelBase``; plus ``project_to_2d`` / ``project_to_2d_linear`` (common/camera.py) and the
data-parallel helpers in ``videopose3d_amd.dp``.

Runtime default set at import (only effective while the HIP runtime of this process has not been initialised yet, i.e. before the
first CUDA/HIP call -- run.py imports the classes at line 21, long before; a value already in the environment wins):
``HIP_FORCE_DEV_KERNARG=1`` -- kernel arguments in device memory instead of host-coherent system memory.  A training step is a
dependent chain of ~110 launches and every kernel start otherwise fetches its arguments across PCIe: measured on MI355X, same box,
alternating processes, six boxes (profiles/r05_dev_kernarg_ab.txt): cfg3 step 4.543 -> 4.409, 4.520 -> 4.394, 4.22 -> 4.07 ms, ...
(-2.6 ... -3.9 %), the eval forward unchanged.  The price is host time per launch (+~2 us: arguments are written through the PCIe BAR), so a HOST-bound step
gets slower when run eagerly -- the 260-launch semi-supervised step 2.05 -> 2.5 ms -- and is unaffected as a hipGraph replay
(graph.GraphedStep: 1.74 ms either way); set ``HIP_FORCE_DEV_KERNARG=0`` for launch-bound eager workloads.
"""
import os as _os

_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from ._lib import Vp3dError, LIB_PATH  # noqa: F401,E402
from .model import TemporalModel, TemporalModelBase, TemporalModelOptimized1f, default_math, set_default_math  # noqa: F401,E402

__all__ = ["TemporalModel", "TemporalModelBase", "TemporalModelOptimized1f", "Vp3dError", "default_math", "set_default_math"]
