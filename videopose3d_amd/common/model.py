"""Import shim with the reference's module path: put ``<repo>/videopose3d_amd`` on ``sys.path`` ahead of the
reference checkout (or copy this file over the reference's ``common/model.py``) and run.py's
``from common.model import *`` (run.py:21) picks up the MI355X classes.  See INTEGRATION.md."""
from videopose3d_amd.model import TemporalModel, TemporalModelBase, TemporalModelOptimized1f  # noqa: F401

__all__ = ["TemporalModelBase", "TemporalModel", "TemporalModelOptimized1f"]
