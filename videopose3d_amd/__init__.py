"""videopose3d_amd -- MI355X (gfx950) native implementation of the VideoPose3D temporal-model hot path.

Public surface = the reference's (common/model.py): ``TemporalModel``, ``TemporalModelOptimized1f``,
``TemporalModelBase``; plus ``project_to_2d`` / ``project_to_2d_linear`` (common/camera.py) and the
data-parallel helpers in ``videopose3d_amd.dp``.
"""
from ._lib import Vp3dError, LIB_PATH  # noqa: F401
from .model import TemporalModel, TemporalModelBase, TemporalModelOptimized1f, default_math, set_default_math  # noqa: F401

__all__ = ["TemporalModel", "TemporalModelBase", "TemporalModelOptimized1f", "Vp3dError", "default_math", "set_default_math"]
