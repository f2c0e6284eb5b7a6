"""The tooling that runs the reference's unmodified run.py (tools/make_synth_h36m.py, tools/runpy_optc_patch.py) against the
reference checkout of the build container.  /root/reference does not exist on the GPU box: these tests skip there."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("VP3D_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "run.py")), reason="no reference checkout on this host")


@pytest.fixture()
def ref_copy(tmp_path):
    """A scratch copy of the reference checkout OUTSIDE the repository (the tools write data/ and run_optc.py next to run.py)."""
    dst = tmp_path / "VideoPose3D"
    shutil.copytree(REF, dst, ignore=shutil.ignore_patterns("__pycache__", "images", ".git"))
    return str(dst)


def test_synthetic_h36m_loads_through_the_reference_dataset_class(ref_copy):
    """make_synth_h36m.py writes the two archives run.py:38-70 reads; the reference's own Human36mDataset / normalisation accept
    them: 7 subjects, 17 joints after remove_static_joints, 4 camera views per action, 2D keypoints inside the image."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synth_h36m.py"), "--reference", ref_copy, "--frames", "60,80"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    sys.path.insert(0, ref_copy)
    try:
        for m in [k for k in sys.modules if k == "common" or k.startswith("common.")]:
            del sys.modules[m]                                   # (the shim's `common` may have been imported by another test)
        from common.h36m_dataset import Human36mDataset
        ds = Human36mDataset(os.path.join(ref_copy, "data", "data_3d_h36m.npz"))
        kp = np.load(os.path.join(ref_copy, "data", "data_2d_h36m_synth.npz"), allow_pickle=True)
        meta, pos2d = kp["metadata"].item(), kp["positions_2d"].item()
        assert sorted(ds.subjects()) == sorted(["S1", "S5", "S6", "S7", "S8", "S9", "S11"])
        assert ds.skeleton().num_joints() == 17 and meta["num_joints"] == 17
        assert meta["keypoints_symmetry"] == [list(ds.skeleton().joints_left()), list(ds.skeleton().joints_right())]
        for s in ds.subjects():
            assert len(ds[s].keys()) == 4                        # 2 actions x 2 takes
            for a in ds[s].keys():
                assert ds[s][a]["positions"].shape[1:] == (17, 3) and len(pos2d[s][a]) == 4
                for cam, v in zip(ds[s][a]["cameras"], pos2d[s][a]):
                    assert v.shape == (ds[s][a]["positions"].shape[0], 17, 2) and v.dtype == np.float32
                    assert np.isfinite(v).all() and (v[..., 0] > -200).all() and (v[..., 0] < cam["res_w"] + 200).all()
    finally:
        sys.path.remove(ref_copy)
        for m in [k for k in sys.modules if k == "common" or k.startswith("common.")]:
            del sys.modules[m]


def test_option_c_patch_applies_to_the_reference_run_py(ref_copy):
    """runpy_optc_patch.py: every textual edit of INTEGRATION.md 3b finds its site in the reference's run.py (16 numpy -> torch
    conversions, the generator / loss imports, the two optimizer constructions), the result compiles, run.py itself is untouched."""
    before = open(os.path.join(ref_copy, "run.py")).read()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "runpy_optc_patch.py"), ref_copy], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "'from_numpy': 16" in r.stdout and "'adam': 2" in r.stdout and "'loss': 1" in r.stdout and "'gen_import': 1" in r.stdout
    assert open(os.path.join(ref_copy, "run.py")).read() == before
    patched = open(os.path.join(ref_copy, "run_optc.py")).read()
    assert "from videopose3d_amd.generators import ChunkedGenerator, UnchunkedGenerator" in patched
    assert "FlatAdam(model_pos_train.parameters(), lr=lr, amsgrad=True)" in patched and "optim.Adam(" not in patched
    assert "torch.from_numpy(batch_3d.astype('float32'))" not in patched
