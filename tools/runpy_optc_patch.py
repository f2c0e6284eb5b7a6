#!/usr/bin/env python3
"""Write `run_optc.py` next to a VideoPose3D checkout's run.py: run.py + the OPT-IN edits of INTEGRATION.md section 3b (device
generators, fused loss, fused Adam), applied textually so that the in-situ effect of each can be measured on the GPU box:

    python tools/runpy_optc_patch.py /path/to/VideoPose3D [generators] [loss] [adam]      (default: all three)

Nothing here is needed for the drop-in itself (run.py runs unmodified behind the import shim); this shows what the three
edits buy once run.py's own Python batch assembly (generators.py:105-149) is what bounds an epoch."""
import os
import re
import sys


def main():
    ref = sys.argv[1]
    what = set(sys.argv[2:]) or {"generators", "loss", "adam"}
    src = open(os.path.join(ref, "run.py")).read()
    n = {}
    if "generators" in what:
        # run.py:23 -- the device generators yield CUDA float32 tensors; the numpy -> torch conversions become pass-throughs
        src, n["from_numpy"] = re.subn(r"torch\.from_numpy\((\w+)\.astype\('float32'\)\)", r"_dev(\1)", src)
        src, n["gen_import"] = re.subn(r"from common\.generators import ChunkedGenerator, UnchunkedGenerator",
                                       "from videopose3d_amd.generators import ChunkedGenerator, UnchunkedGenerator\n"
                                       "_dev = lambda a: a if torch.is_tensor(a) else torch.from_numpy(a.astype('float32'))", src)
        assert n["gen_import"] == 1 and n["from_numpy"] == 16, n
    if "loss" in what:
        # run.py:22 -- mpjpe / weighted_mpjpe on the HIP path (one kernel each: value + gradient); the protocol metrics stay
        src, n["loss"] = re.subn(r"from common\.loss import \*", "from common.loss import *\nfrom videopose3d_amd.loss import mpjpe, weighted_mpjpe",
                                 src)
        assert n["loss"] == 1, n
    if "adam" in what:
        # run.py:252,264 -- same constructor arguments, same state_dict layout
        src, n["adam"] = re.subn(r"optim\.Adam\(", "FlatAdam(", src)
        src = src.replace("import torch.optim as optim", "import torch.optim as optim\nfrom videopose3d_amd.optim import FlatAdam")
        assert n["adam"] == 2, n
    out = os.path.join(ref, "run_optc.py")
    with open(out, "w") as f:
        f.write(src)
    compile(src, out, "exec")
    print("wrote %s: %s" % (out, n))


if __name__ == "__main__":
    main()
