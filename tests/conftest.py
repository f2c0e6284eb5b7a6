import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    config.addinivalue_line("markers", "single_arithmetic: the test does not depend on the GEMM arithmetic (fp32-only entry "
                            "points, index / byte kernels): it runs once, under the f32 id of the math_mode fixture")
    config.addinivalue_line("markers", "allow_f32_engine: the f16x3 run of this test may also touch the fp32 engine")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no GPU is visible, e.g. a plain `pytest tests/` here."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.hookimpl(tryfirst=True, hookwrapper=True)
def pytest_runtest_makereport(item, call):
    """Keeps each phase's report on the item (item.rep_call ...) so that fixtures can see how the test body ended."""
    outcome = yield
    rep = outcome.get_result()
    setattr(item, "rep_" + rep.when, rep)


@pytest.fixture(params=["f16x3", "f32"])
def math_mode(request):
    """Model-level GPU tests run once per GEMM arithmetic: "f16x3" (split-fp16 MFMA, engine_s16) and "f32" (fp32 MFMA) --
    same oracle, same tolerances.  No silent duplicates: the engine counts which GEMM path served every call
    (engine.ENGINE_CALLS) and the "f16x3" run of a test must have executed the split-fp16 engine and nothing else.  Tests
    that do not depend on the arithmetic carry @pytest.mark.single_arithmetic and run under the f32 id only;
    configurations engine_s16.supported() rejects skip themselves (tests/util.s16_or_skip)."""
    import videopose3d_amd as _V
    from videopose3d_amd import engine as _E
    single = request.node.get_closest_marker("single_arithmetic") is not None
    if request.param == "f16x3" and single:
        pytest.skip("independent of the GEMM arithmetic: runs once, under the f32 id")
    keep = dict(_E.S16_MIN_FORWARD_FLOPS)
    _E.S16_MIN_FORWARD_FLOPS.update({True: 0.0, False: 0.0})     # the small test models must not fall below the
    _V.set_default_math(request.param)                            # engine's "big enough to be compute-bound" threshold
    _E.ENGINE_CALLS.clear()
    yield request.param
    calls = dict(_E.ENGINE_CALLS)
    _V.set_default_math(None)
    _E.S16_MIN_FORWARD_FLOPS.update(keep)
    rep = getattr(request.node, "rep_call", None)
    if single or rep is None or not rep.passed:       # skipped / failed bodies have nothing to certify
        return
    s16 = calls.get("s16_train", 0) + calls.get("s16_eval", 0)
    f32 = calls.get("f32_train", 0) + calls.get("f32_eval", 0)
    if request.param == "f32":
        assert s16 == 0, calls
    else:
        assert s16 > 0, "this test claims f16x3 coverage but never ran the split-fp16 engine: %r" % (calls,)
        if not request.node.get_closest_marker("allow_f32_engine"):
            assert f32 == 0, "part of this test fell back to the fp32 engine under the f16x3 id: %r" % (calls,)
