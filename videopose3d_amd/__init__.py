"""videopose3d_amd -- MI355X (gfx950) native implementation of the VideoPose3D temporal-model hot path.

Public surface = the reference's (common/model.py): ``TemporalModel``, ``TemporalModelOptimized1f``,
``TemporalModelBase``; plus ``project_to_2d`` / ``project_to_2d_linear`` (common/camera.py) and the
data-parallel helpers in ``videopose3d_amd.dp``.

Nothing in the process environment is changed by importing the package (round 6; rounds 5's import-time default is gone).
``HIP_FORCE_DEV_KERNARG=1`` -- a ROCm runtime variable: kernel arguments in device memory instead of host-coherent system
memory -- is worth -2.6 ... -3.9 % on the GPU-bound training step (a dependent chain of ~200 launches, each of which otherwise
fetches its arguments across PCIe; profiles/r05_dev_kernarg_ab.txt, six boxes) and COSTS a host-bound eager workload ~2 us per
launch (the 260-launch semi-supervised step: 2.05 -> 2.5 ms eagerly, unchanged as a hipGraph replay).  It is therefore the
caller's decision: export it in the launcher (``bench.py`` does), or opt in with ``VP3D_DEV_KERNARG=1`` /
``videopose3d_amd.enable_device_kernargs()`` BEFORE the process's first HIP call -- both say so once when they come too late.
"""
import os as _os
import warnings as _warnings


def enable_device_kernargs() -> bool:
    """Put ``HIP_FORCE_DEV_KERNARG=1`` into this process's environment (a value already there wins).  Only effective while
    the HIP runtime has not been initialised: returns False -- and warns once -- when it already has."""
    import torch as _torch
    if _torch.cuda.is_initialized():
        _warnings.warn("videopose3d_amd.enable_device_kernargs(): the HIP runtime of this process is already initialised; "
                       "HIP_FORCE_DEV_KERNARG has no effect now -- export it before starting Python", RuntimeWarning, stacklevel=2)
        return False
    _os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    return _os.environ["HIP_FORCE_DEV_KERNARG"] == "1"


if _os.environ.get("VP3D_DEV_KERNARG", "0") == "1":
    enable_device_kernargs()

from ._lib import Vp3dError, LIB_PATH  # noqa: F401,E402
from .model import TemporalModel, TemporalModelBase, TemporalModelOptimized1f, default_math, set_default_math  # noqa: F401,E402

__all__ = ["TemporalModel", "TemporalModelBase", "TemporalModelOptimized1f", "Vp3dError", "default_math", "set_default_math",
           "enable_device_kernargs"]
