"""A/B helper: run a script of this repo against another build of the HIP library.
    python tools/ab_lib.py videopose3d_amd/libvp3d_b.so bench.py --steps 10 ..."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import videopose3d_amd._lib as L  # noqa: E402

L.LIB_PATH = os.path.abspath(sys.argv[1])
script = sys.argv[2]
sys.argv = sys.argv[2:]
runpy.run_path(script, run_name="__main__")
