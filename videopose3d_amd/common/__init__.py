"""Shadow package for the reference's ``common``: only ``common.model`` is replaced.  When this directory is
placed on ``sys.path`` ahead of a VideoPose3D checkout, every other ``common.*`` module (camera, loss,
generators, ...) must still resolve to the reference, so the reference's ``common`` directory (found by walking
``sys.path``) is appended to this package's ``__path__``."""
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
for _p in list(sys.path):
    _cand = os.path.join(_p or ".", "common")
    if os.path.isdir(_cand) and os.path.abspath(_cand) != _here and os.path.exists(os.path.join(_cand, "generators.py")):
        if _cand not in __path__:
            __path__.append(_cand)
        break
