"""ctypes binding of libvp3d.so (C ABI declared in include/vp3d.h).

The library is hand-written HIP for gfx950 and is built in-tree by ``__graft_entry__.build()`` (``python __graft_entry__.py``).  There is NO fallback: if the shared object is missing or a call
fails, an exception is raised -- the product path never silently routes around the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvp3d.so")


class Vp3dError(RuntimeError):
    pass


class RowMap(C.Structure):
    _fields_ = [("batch", C.c_int32), ("t_dst", C.c_int32), ("t_src", C.c_int32), ("t_stride", C.c_int32),
                ("tap_step", C.c_int32), ("t_off", C.c_int32), ("taps", C.c_int32)]


class Dropout(C.Structure):
    _fields_ = [("p", C.c_float), ("seed", C.c_uint64), ("offset", C.c_uint64), ("layer", C.c_uint32),
                ("offset_ptr", C.c_void_p)]


class ActBwd(C.Structure):
    _fields_ = [("y_up", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("mean", C.c_void_p),
                ("invstd", C.c_void_p), ("drop", C.POINTER(Dropout)), ("g_out", C.c_void_p), ("partials", C.c_void_p),
                ("c_stat", C.c_int32), ("store_v", C.c_int32)]


class Epilogue(C.Structure):
    _fields_ = [("bias", C.c_void_p), ("relu", C.c_int32), ("residual", C.c_void_p), ("r_bpitch", C.c_int64),
                ("r_ld", C.c_int32), ("r_t", C.c_int32), ("r_stride", C.c_int32), ("r_off", C.c_int32),
                ("r_col0", C.c_int32), ("r_cols", C.c_int32), ("stat_sum", C.c_void_p), ("stat_m2", C.c_void_p),
                ("act_bwd", C.POINTER(ActBwd))]


class S16Opts(C.Structure):
    _fields_ = [("x_bound", C.c_void_p), ("w_bound", C.c_void_p), ("amax_out", C.c_void_p), ("cfg", C.c_int32),
                ("splits", C.c_int32), ("ws", C.c_void_p), ("ws_floats", C.c_int64), ("raw_partials", C.c_int32),
                ("res_s16", C.c_int32), ("res_bound", C.c_void_p), ("out_s16", C.c_int32), ("in_amax", C.c_void_p),
                ("l1", C.c_void_p), ("res_amax", C.c_void_p), ("out_wbound", C.c_void_p), ("no_output", C.c_int32),
                ("act_scale", C.c_void_p), ("act_shift", C.c_void_p), ("act_drop", C.POINTER(Dropout)),
                ("act_bound", C.c_void_p), ("act_bits", C.c_void_p), ("tickets", C.c_void_p), ("red", C.c_void_p),
                ("stat_slab_rows", C.c_int32), ("fin", C.c_void_p)]


class S16Fin(C.Structure):                     # include/vp3d.h: vp3d_s16_fin
    _fields_ = [("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float), ("momentum", C.c_float),
                ("momentum_dev", C.c_void_p), ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("num_batches_tracked", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("save_mean", C.c_void_p),
                ("save_invstd", C.c_void_p), ("tickets", C.c_void_p)]


class S16Red(C.Structure):                     # include/vp3d.h: vp3d_s16_red
    _fields_ = [("y_up", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p), ("scale", C.c_void_p),
                ("act_bits", C.c_void_p), ("rows_up", C.c_int64), ("c_up", C.c_int32), ("p", C.c_float),
                ("partials", C.c_void_p), ("partials_floats", C.c_int64), ("tickets", C.c_void_p), ("dgamma", C.c_void_p),
                ("dbeta", C.c_void_p), ("dy_bound", C.c_void_p)]


class Gather(C.Structure):
    _fields_ = [("n_chunks", C.c_int32), ("chunks", C.c_void_p), ("seq_off", C.c_void_p), ("poses_2d", C.c_void_p),
                ("j2", C.c_int32), ("f2", C.c_int32), ("kps_perm", C.c_void_p), ("poses_3d", C.c_void_p),
                ("j3", C.c_int32), ("f3", C.c_int32), ("joints_perm", C.c_void_p), ("cameras", C.c_void_p),
                ("cam_dim", C.c_int32), ("chunk_length", C.c_int32), ("pad", C.c_int32), ("causal_shift", C.c_int32),
                ("out_2d", C.c_void_p), ("out_3d", C.c_void_p), ("out_cam", C.c_void_p)]


class Adam(C.Structure):
    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("weight_decay", C.c_double), ("step", C.c_int64), ("amsgrad", C.c_int32)]


_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
_P = C.POINTER

# name -> (restype, argtypes); every symbol declared in include/vp3d.h must appear here
SIGNATURES = {
    "vp3d_version": (C.c_int, []),
    "vp3d_launch_count": (_i64, []),
    "vp3d_last_error": (C.c_char_p, []),
    "vp3d_stat_slabs": (_i64, [_i64]),
    "vp3d_rows_gemm_splits": (C.c_int, [_i64, _i32, _i32]),
    "vp3d_rows_gemm_ws_floats": (_i64, [_i64, _i32, _i32]),
    "vp3d_wgrad_splits": (C.c_int, [_i64, _i32, _i32]),
    "vp3d_tconv_fwd": (C.c_int, [_vp, _P(RowMap), _vp, _i32, _i32, _vp, _i32, _i32, _vp, _i64, _i32, _P(Epilogue), _vp,
                                 _vp, _i64]),
    "vp3d_nt_s16_plan": (C.c_int, [_i64, _i32, _i32, _i32, _P(_i32), _P(_i32)]),
    "vp3d_nt_s16_stat_slab_rows": (C.c_int, [_i32]),
    "vp3d_bn_finalize_slab": (C.c_int, [_vp, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vp3d_im2row_split_s16": (C.c_int, [_vp, _P(RowMap), _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i64]),
    "vp3d_amax_floor": (C.c_int, [_vp, _i64, _vp, _f32, _vp]),
    "vp3d_expand_fwd_s16": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _P(Dropout), _vp, _vp, _vp]),
    "vp3d_expand_bwd_p_s16": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _f32, _vp, _i64, _vp, _vp, _vp, _P(_i32)]),
    "vp3d_prologue_a_s16": (C.c_int, [_vp, _i32, _P(_vp), _P(_i64), _P(_vp), _P(_f32), _i32, _i32, _P(_vp), _P(_vp), _P(_i64),
                                      _P(_i32), _f32, _vp]),
    "vp3d_prologue_b_s16": (C.c_int, [_vp, _P(RowMap), _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _vp,
                                      _vp, _vp, _i32, _P(_vp), _P(_i32), _i32, _i32, _vp, _P(_vp), _P(_vp)]),
    "vp3d_has_experiments": (C.c_int, []),
    "vp3d_nt_s16_workspace": (C.c_int, [_i64, _i32, _i32, _i32, _i32, _i32, _P(_i64), _P(_i32)]),
    "vp3d_tconv_nt_s16": (C.c_int, [_vp, _P(RowMap), _vp, _i32, _i32, _vp, _i32, _i32, _vp, _i64, _i32, _P(Epilogue), _vp,
                                    _P(S16Opts)]),
    "vp3d_split_rows": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _vp, _i64, _vp]),
    "vp3d_amax": (C.c_int, [_vp, _i64, _vp, _vp]),
    "vp3d_bn_act_fwd_s16": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _P(Dropout), _vp, _vp, _i32, _i32, _i32, _i32, _i32,
                                      _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "vp3d_bn_bwd_apply_s16": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _P(Dropout), _vp, _vp, _vp, _vp, _vp,
                                        _vp, _i64]),
    "vp3d_wgrad_rows_s16": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _i64, _i32, _i32, _vp, _i32, _vp]),
    "vp3d_bn_bwd_reduce_bits": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _P(_i32)]),
    "vp3d_bn_bwd_reduce_fin_s16": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                             _vp, _P(_i32), _P(_i32), _P(_i32)]),
    "vp3d_act_mask_s16": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _i64]),
    "vp3d_gather_t_s16": (C.c_int, [_vp, _P(RowMap), _vp, _i32, _vp, _i64]),
    "vp3d_sum_slices": (C.c_int, [_vp, _i64, _i32, _vp, _vp]),
    "vp3d_expand_bwd_s16": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                      _vp]),
    "vp3d_expand_bwd_gram_s16": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i64, _i32, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp,
                                           _vp, _vp, _vp, _vp]),
    "vp3d_split_t": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _vp, _vp, _i64, _vp, _i64]),
    "vp3d_pack_weight_s16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _i64, _vp, _i64, _i32]),
    "vp3d_amax_multi": (C.c_int, [_vp, _i32, _P(_vp), _P(_i64), _vp]),
    "vp3d_pack_weight_s16_multi": (C.c_int, [_vp, _i32, _P(_vp), _P(_i32), _i32, _i32, _vp, _P(_vp), _P(_vp)]),
    "vp3d_act_bounds_multi": (C.c_int, [_vp, _i32, _i32, _P(_vp), _P(_vp), _P(_i64), _P(_i32), _f32, _vp]),
    "vp3d_bn_bwd_finalize_s16": (C.c_int, [_vp, _i32, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _f32, _vp]),
    "vp3d_act_bound": (C.c_int, [_vp, _i32, _i64, _vp, _vp, _f32, _vp, _vp]),
    "vp3d_dy_bound": (C.c_int, [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _f32, _vp]),
    "vp3d_tconv_dgrad": (C.c_int, [_vp, _P(RowMap), _vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp, _i64, _i32,
                                   _P(Epilogue), _vp, _vp, _i64]),
    "vp3d_tconv_wgrad": (C.c_int, [_vp, _P(RowMap), _vp, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _vp]),
    "vp3d_wgrad_reduce": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vp3d_pack_weight": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _i32]),
    "vp3d_im2row": (C.c_int, [_vp, _P(RowMap), _vp, _i32, _i32, _i32, _i32, _vp]),
    "vp3d_bn_fold": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _f32, _vp, _vp]),
    "vp3d_bn_finalize": (C.c_int, [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vp3d_bn_finalize_dm": (C.c_int, [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vp3d_bn_act_fwd": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _P(Dropout), _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vp3d_bn_bwd_reduce": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _P(Dropout), _vp, _P(_i32)]),
    "vp3d_bn_bwd_finalize": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp]),
    "vp3d_bn_bwd_apply": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _P(Dropout), _vp, _vp, _vp]),
    "vp3d_bn_bwd_apply_g": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vp3d_act_bwd_parts": (_i64, [_i64, _i32, _i32]),
    "vp3d_colsum": (C.c_int, [_vp, _i64, _i32, _vp, _i32, _vp]),
    "vp3d_head_supported": (C.c_int, [_i64, _i32, _i32]),
    "vp3d_head_bwd_ws_floats": (_i64, [_i64, _i32, _i32]),
    "vp3d_head_fwd": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    "vp3d_head_bwd": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vp3d_head_fold": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    "vp3d_dropout_mask": (C.c_int, [_vp, _i64, _P(Dropout), _vp]),
    "vp3d_project_to_2d_fwd": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _i32, _vp]),
    "vp3d_project_to_2d_bwd": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _i32, _vp]),
    "vp3d_gather_chunks": (C.c_int, [_vp, _P(Gather)]),
    "vp3d_tta_fold": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    "vp3d_mpjpe_ws_bytes": (_i64, [_i64]),
    "vp3d_mpjpe": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vp3d_adam_step": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _P(Adam)]),
    "vp3d_expand_stats_gram_groups": (C.c_int, [_i64]),
    "vp3d_expand_stats_gram_s16": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _vp,
                                             _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vp3d_range_cols": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _vp, _vp]),
    "vp3d_range_max_tensors": (C.c_int, []),
    "vp3d_range_stats": (C.c_int, [_vp, _i32, _i32, _P(_vp), _P(_vp), _P(_f32), _i32, _P(_vp), _P(_i64), _P(_i64), _vp, _i64, _vp]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle.  Raises Vp3dError when the HIP library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Vp3dError(
            "libvp3d.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU / PyTorch fallback for this path." % LIB_PATH)
    # torch must be imported first so that its bundled libamdhip64.so.7 (same SONAME as /opt/rocm's) is the one
    # HIP runtime in the process: torch's stream handles are then valid inside our launches.
    import torch  # noqa: F401
    h = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(h, name)
        except AttributeError as e:  # pragma: no cover
            raise Vp3dError("libvp3d.so does not export %s (stale build?)" % name) from e
        fn.restype = res
        fn.argtypes = args
    if h.vp3d_version() != 110:
        raise Vp3dError("libvp3d.so version %d does not match the Python host (110); rebuild" % h.vp3d_version())
    _lib = h
    return h


def check(rc, what=""):
    if rc != 0:
        msg = lib().vp3d_last_error()
        raise Vp3dError("%s failed (code %d): %s" % (what or "vp3d call", rc, msg.decode() if msg else "?"))
