"""ctypes binding of libmvfnet_hip.so (the C ABI in include/mvfnet_hip.h).

The library is built in-tree by `python -m mvfnet_amd.build` (or __graft_entry__.build()).
There is NO fallback: if the shared object is missing or fails to load, importing this module
raises, and every op in mvfnet_amd.ops raises with it.
"""
import ctypes as C
import os

import torch  # noqa: F401  -- FIRST: torch bundles its own libamdhip64; ours must resolve to that same runtime

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MVF_LIB_PATH") or os.path.join(_HERE, "libmvfnet_hip.so")      # (override: tooling only, e.g. timing-ablation builds)

MVF_F32, MVF_BF16 = 0, 1
MVF_NCHW, MVF_NHWC = 0, 1
VIEW_T, VIEW_H, VIEW_W = 1, 2, 4
MODE_BITS = {"T": 1, "TH": 3, "THW": 7}
ERRNAMES = {-1: "MVF_EINVAL", -2: "MVF_ESHAPE", -3: "MVF_EWS", -4: "MVF_EHIP", -5: "MVF_EUNSUPPORTED"}


class MvfDesc(C.Structure):
    _fields_ = [("nt", C.c_int32), ("c", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("n_segment", C.c_int32),
                ("cs", C.c_int32), ("mode", C.c_int32), ("layout", C.c_int32), ("dtype", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32),
                ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("ho", C.c_int32), ("wo", C.c_int32), ("x_pix_stride", C.c_int32),
                ("dtype", C.c_int32), ("relu", C.c_int32), ("split_c", C.c_int32), ("x2_pix_stride", C.c_int32), ("in_dil", C.c_int32),
                ("res_c0", C.c_int32), ("x_c0", C.c_int32)]


class PackJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("out", C.c_void_p), ("cout", C.c_int32), ("cin", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
                ("kw_pad", C.c_int32), ("cin_pad", C.c_int32), ("kind", C.c_int32), ("first_block", C.c_int32)]


class PlanOp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_int", C.c_int32), ("n_flt", C.c_int32), ("reserved", C.c_int32), ("fn", C.c_void_p), ("word0", C.c_longlong),
                ("float0", C.c_longlong)]


class SgdSegment(C.Structure):
    _fields_ = [("first", C.c_int64), ("lr_mult", C.c_float), ("decay_mult", C.c_float)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "mvfnet_amd: %s not found. Build it with `python -m mvfnet_amd.build` (needs hipcc, "
            "--offload-arch=gfx950). There is no CPU or PyTorch fallback for the hot path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, fp, sz, i32, f32 = C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_float
    dp = C.POINTER(MvfDesc)
    lib.mvf_abi_version.restype = i32
    lib.mvf_abi_version.argtypes = []
    lib.mvf_last_error.restype = C.c_char_p
    lib.mvf_last_error.argtypes = []
    lib.mvf_fwd_infer.restype = i32
    lib.mvf_fwd_infer.argtypes = [dp, vp, vp, fp, fp, fp, fp, fp, vp]
    lib.mvf_fwd_train_workspace_bytes.restype = sz
    lib.mvf_fwd_train_workspace_bytes.argtypes = [dp]
    lib.mvf_fwd_train.restype = i32
    lib.mvf_fwd_train.argtypes = [dp, vp, vp, fp, fp, fp, fp, fp, f32, f32, fp, fp, fp, fp, vp, sz, vp]
    lib.mvf_bwd_workspace_bytes.restype = sz
    lib.mvf_bwd_workspace_bytes.argtypes = [dp]
    lib.mvf_bwd.restype = i32
    lib.mvf_bwd.argtypes = [dp, vp, vp, fp, fp, fp, fp, fp, fp, fp, i32, vp, fp, fp, fp, fp, fp, vp, sz, vp]
    lib.mvf_fwd_infer_slice.restype = i32
    lib.mvf_fwd_infer_slice.argtypes = [dp, vp, vp, fp, fp, fp, fp, fp, vp]
    cp = C.POINTER(ConvDesc)
    lib.mvf_conv2d_nhwc_fwd.restype = i32
    lib.mvf_conv2d_nhwc_fwd.argtypes = [cp, vp, vp, vp, fp, vp, vp, vp]
    lib.mvf_conv2d_workspace_bytes.restype = sz
    lib.mvf_conv2d_workspace_bytes.argtypes = [cp]
    lib.mvf_conv2d_nhwc_fwd_ws.restype = i32
    lib.mvf_conv2d_nhwc_fwd_ws.argtypes = [cp, vp, vp, vp, fp, vp, vp, vp, sz, vp]
    lib.mvf_conv2d_nhwc_dgrad_bnsums.restype = i32
    lib.mvf_conv2d_nhwc_dgrad_bnsums.argtypes = [cp, vp, vp, vp, vp, fp, fp, fp, fp, fp, vp, sz, vp]
    lib.mvf_bn_bwd_finalize.restype = i32
    lib.mvf_bn_bwd_finalize.argtypes = [fp, i32, i32, fp, fp, vp]
    lib.mvf_conv2d_nhwc_fwd_bnapply.restype = i32
    lib.mvf_conv2d_nhwc_fwd_bnapply.argtypes = [cp, vp, vp, vp, fp, fp, vp, fp, fp, vp, vp, vp, sz, vp]
    lib.mvf_conv2d_nhwc_fwd_bnbwd_sums.restype = i32
    lib.mvf_conv2d_nhwc_fwd_bnbwd_sums.argtypes = [cp, vp, vp, vp, vp, vp, fp, fp, fp, vp, sz, vp]
    lib.mvf_conv2d_nhwc_fwd_bnbwd_apply.restype = i32
    lib.mvf_conv2d_nhwc_fwd_bnbwd_apply.argtypes = [cp, vp, vp, vp, vp, vp, fp, fp, fp, fp, fp, vp, vp, sz, vp]
    lib.mvf_conv2d_nhwc_fwd_resmask.restype = i32
    lib.mvf_conv2d_nhwc_fwd_resmask.argtypes = [cp, vp, vp, vp, fp, vp, vp, vp, vp, sz, vp]
    i64_ = C.c_long
    lib.mvf_conv2d_nhwc_fwd_resmask_gate.restype = i32
    lib.mvf_conv2d_nhwc_fwd_resmask_gate.argtypes = [cp, vp, vp, vp, fp, vp, vp, vp, vp, vp, sz, vp]
    lib.mvf_conv2d_nhwc_fwd_resmask_gate_colsums.restype = i32
    lib.mvf_conv2d_nhwc_fwd_resmask_gate_colsums.argtypes = [cp, vp, vp, vp, vp, vp, vp, vp, fp, vp, sz, vp]
    lib.mvf_nhwc_stencil_gate_colsums.restype = i32
    lib.mvf_nhwc_stencil_gate_colsums.argtypes = [dp, vp, i32, vp, i32, fp, fp, fp, i32, vp, i32, vp, vp, fp, vp]
    lib.mvf_conv2d_nhwc_dgrad_bnsums_split.restype = i32
    lib.mvf_conv2d_nhwc_dgrad_bnsums_split.argtypes = [cp, vp, vp, vp, fp, vp, vp, fp, fp, fp, fp, fp, vp, sz, vp]
    lib.mvf_bn_bwd_dzfree_prep.restype = i32
    lib.mvf_bn_bwd_dzfree_prep.argtypes = [vp, i32, i32, fp, fp, fp, fp, fp, i64_, vp, fp, i32, vp]
    lib.mvf_bn_bwd_dzfree_wgrad.restype = i32
    lib.mvf_bn_bwd_dzfree_wgrad.argtypes = [fp, vp, fp, fp, fp, fp, fp, fp, fp, i64_, i32, i32, i32, vp]
    lib.mvf_bn_bwd_dzfree_sums.restype = i32
    lib.mvf_bn_bwd_dzfree_sums.argtypes = [fp, vp, i32, i32, fp, fp, fp, i32, i32, fp, i32, fp, fp, i32, vp]
    lib.mvf_nhwc_stencil_stats_rows.restype = i32
    lib.mvf_nhwc_stencil_stats_rows.argtypes = [dp, i32, i32]
    lib.mvf_nhwc_stencil_tile_plan.restype = i32
    lib.mvf_nhwc_stencil_tile_plan.argtypes = [dp, i32, i32, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.mvf_nhwc_stencil_stats.restype = i32
    lib.mvf_nhwc_stencil_stats.argtypes = [dp, vp, i32, vp, i32, fp, fp, fp, fp, fp, vp]
    lib.mvf_nhwc_stencil_gate.restype = i32
    lib.mvf_nhwc_stencil_gate.argtypes = [dp, vp, i32, vp, i32, fp, fp, fp, fp, fp, i32, vp, i32, vp, vp, vp]
    lib.mvf_conv2d_stats_rows.restype = i32
    lib.mvf_conv2d_stats_rows.argtypes = [cp]
    lib.mvf_conv2d_nhwc_fwd_stats.restype = i32
    lib.mvf_conv2d_nhwc_fwd_stats.argtypes = [cp, vp, vp, vp, vp, fp, fp, vp, sz, vp]
    lib.mvf_pack_conv_weight.restype = i32
    lib.mvf_pack_conv_weight.argtypes = [fp, i32, i32, i32, i32, i32, i32, fp, vp, i32, vp]
    lib.mvf_pack_conv_weights_batched.restype = i32
    lib.mvf_pack_conv_weights_batched.argtypes = [vp, i32, i32, i32, vp]
    lib.mvf_bn_fold.restype = i32
    lib.mvf_bn_fold.argtypes = [fp, fp, fp, fp, f32, i32, fp, fp, vp]
    lib.mvf_stem_prep.restype = i32
    lib.mvf_stem_prep.argtypes = [fp, i32, i32, i32, i32, i32, i32, vp, i32, vp]
    lib.mvf_frames_prep_u8.restype = i32
    lib.mvf_frames_prep_u8.argtypes = [vp, i32, i32, i32, vp, i32, i32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                       i32, i32, i32, i32, vp, vp, i32, vp]
    lib.mvf_maxpool3x3s2_nhwc.restype = i32
    lib.mvf_maxpool3x3s2_nhwc.argtypes = [vp, i32, i32, i32, i32, vp, i32, vp]
    lib.mvf_head_pool_fc.restype = i32
    lib.mvf_head_pool_fc.argtypes = [vp, i32, i32, i32, i32, fp, fp, i32, fp, fp, i32, vp]
    lib.mvf_average_clip.restype = i32
    lib.mvf_average_clip.argtypes = [fp, i32, i32, i32, fp, vp]
    i64, ll = C.c_long, C.c_void_p
    lib.mvf_bn_workspace_bytes.restype = sz
    lib.mvf_bn_workspace_bytes.argtypes = [i64, i32]
    lib.mvf_bn_train_stats.restype = i32
    lib.mvf_bn_train_stats.argtypes = [vp, i64, i32, fp, fp, f32, f32, fp, fp, fp, fp, fp, fp, vp, sz, i32, vp]
    lib.mvf_bn_train_finalize.restype = i32
    lib.mvf_bn_train_finalize.argtypes = [fp, i32, i64, i32, fp, fp, f32, f32, fp, fp, fp, fp, fp, fp, vp]
    lib.mvf_bn_train_stats_gram.restype = i32
    lib.mvf_bn_train_stats_gram.argtypes = [fp, fp, vp, i64, i32, i32, fp, fp, f32, f32, fp, fp, fp, fp, fp, fp, i32, vp]
    lib.mvf_bn_apply_colmeans.restype = i32
    lib.mvf_bn_apply_colmeans.argtypes = [vp, i64, i32, fp, fp, i32, vp, fp, vp, sz, i32, vp]
    lib.mvf_bn_apply.restype = i32
    lib.mvf_bn_apply.argtypes = [vp, i64, i32, fp, fp, vp, fp, fp, i32, vp, i32, vp]
    lib.mvf_bn_apply_bits.restype = i32
    lib.mvf_bn_apply_bits.argtypes = [vp, i64, i32, fp, fp, vp, fp, fp, i32, vp, vp, i32, vp]
    lib.mvf_bn_bwd_apply_masked.restype = i32
    lib.mvf_bn_bwd_apply_masked.argtypes = [vp, i32, vp, vp, i64, i32, fp, fp, fp, fp, fp, fp, fp, i32, vp, i32, vp]
    lib.mvf_bn_bwd_reduce.restype = i32
    lib.mvf_bn_bwd_reduce.argtypes = [vp, i32, vp, vp, i64, i32, fp, fp, fp, fp, i32, vp, fp, fp, vp, sz, i32, vp]
    lib.mvf_bn_bwd_apply.restype = i32
    lib.mvf_bn_bwd_apply.argtypes = [vp, i32, vp, i64, i32, fp, fp, fp, fp, fp, fp, fp, i32, vp, i32, vp]
    lib.mvf_maxpool_bn_relu_fwd.restype = i32
    lib.mvf_maxpool_bn_relu_fwd.argtypes = [vp, i32, i32, i32, i32, fp, fp, vp, vp, i32, vp]
    lib.mvf_maxpool_bn_relu_bwd.restype = i32
    lib.mvf_maxpool_bn_relu_bwd.argtypes = [vp, vp, i32, i32, i32, i32, vp, i32, vp]
    lib.mvf_maxpool_bwd_sums_rows.restype = i32
    lib.mvf_maxpool_bwd_sums_rows.argtypes = [i32, i32]
    lib.mvf_maxpool_bn_relu_bwd_apply.restype = i32
    lib.mvf_maxpool_bn_relu_bwd_apply.argtypes = [vp, vp, i32, i32, i32, i32, vp, fp, fp, fp, fp, fp, fp, fp, vp, i32, vp]
    lib.mvf_maxpool_bn_relu_bwd_sums.restype = i32
    lib.mvf_maxpool_bn_relu_bwd_sums.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, fp, fp, fp, fp, fp, i32, vp]
    lib.mvf_head_train_fwd.restype = i32
    lib.mvf_head_train_fwd.argtypes = [vp, i32, i32, i32, i32, fp, fp, i32, ll, fp, fp, fp, fp, fp, fp, i32, vp]
    lib.mvf_conv2d_nhwc_fwd_mvf.restype = i32
    lib.mvf_conv2d_nhwc_fwd_mvf.argtypes = [cp, vp, vp, fp, fp, i32, i32, i32, vp, vp, sz, vp]
    lib.mvf_bn_bwd_pair.restype = i32
    lib.mvf_bn_bwd_pair.argtypes = [vp, i32, vp, vp, vp, i64, i32, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, vp, vp, vp, sz, i32, vp]
    lib.mvf_bn_bwd_wgrad_splits.restype = i32
    lib.mvf_bn_bwd_wgrad_splits.argtypes = [i64, i32, i32, i32, i32]
    lib.mvf_bn_bwd_wgrad_slab_bytes.restype = sz
    lib.mvf_bn_bwd_wgrad_slab_bytes.argtypes = [i64, i32, i32, i32, i32]
    lib.mvf_bn_bwd_apply_wgrad.restype = i32
    lib.mvf_bn_bwd_apply_wgrad.argtypes = [vp, i32, vp, vp, i64, i32, fp, fp, fp, fp, fp, fp, fp, i32, vp, vp, i32, i32, fp, sz, i32, vp]
    lib.mvf_bn_bwd_pair_wgrad.restype = i32
    lib.mvf_bn_bwd_pair_wgrad.argtypes = [vp, i32, vp, vp, vp, i64, i32, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, vp, vp, vp, i32, vp, i32, i32,
                                          fp, fp, sz, vp, sz, i32, vp]
    lib.mvf_conv1x1_bwd_fused_splits.restype = i32
    lib.mvf_conv1x1_bwd_fused_splits.argtypes = [i64, i32, i32]
    lib.mvf_conv1x1_bwd_fused.restype = i32
    lib.mvf_conv1x1_bwd_fused.argtypes = [vp, i32, vp, vp, i32, vp, i64, i32, i32, fp, fp, fp, fp, fp, vp, fp, fp, fp, fp, vp, fp, i32, fp, sz, i32, vp]
    lib.mvf_conv1x1_bnbwd_sums_pair.restype = i32
    lib.mvf_conv1x1_bnbwd_sums_pair.argtypes = [vp, i32, vp, vp, i32, vp, vp, i32, vp, i64, i32, i32, fp, fp, fp, fp, fp, fp, i32, i32, vp]
    lib.mvf_wgrad_slab_reduce.restype = i32
    lib.mvf_wgrad_slab_reduce.argtypes = [fp, i32, i32, i32, fp, vp]
    lib.mvf_ce_loss.restype = i32
    lib.mvf_ce_loss.argtypes = [fp, ll, i32, i32, fp, fp, fp, vp]
    lib.mvf_head_train_bwd.restype = i32
    lib.mvf_head_train_bwd.argtypes = [fp, fp, fp, fp, i32, i32, i32, i32, i32, fp, fp, fp, vp, i32, vp]
    lib.mvf_conv2d_wgrad_workspace_bytes.restype = sz
    lib.mvf_conv2d_wgrad_workspace_bytes.argtypes = [cp]
    lib.mvf_conv2d_nhwc_wgrad.restype = i32
    lib.mvf_conv2d_nhwc_wgrad.argtypes = [cp, vp, vp, vp, i32, i32, i32, i32, fp, vp, sz, vp]
    lib.mvf_conv2d_nhwc_wgrad_wgs.restype = i32
    lib.mvf_conv2d_nhwc_wgrad_wgs.argtypes = [cp, vp, vp, vp, i32, i32, i32, i32, fp, vp, sz, i32, vp]
    lib.mvf_pack_conv_weight_dgrad.restype = i32
    lib.mvf_pack_conv_weight_dgrad.argtypes = [fp, i32, i32, i32, i32, vp, i32, vp]
    lib.mvf_nhwc_stencil.restype = i32
    lib.mvf_nhwc_stencil.argtypes = [dp, vp, i32, vp, i32, fp, fp, fp, fp, fp, i32, vp, i32, vp, vp]
    lib.mvf_nhwc_tapgrad_workspace_bytes.restype = sz
    lib.mvf_nhwc_tapgrad_workspace_bytes.argtypes = [dp]
    lib.mvf_nhwc_tapgrad.restype = i32
    lib.mvf_nhwc_tapgrad.argtypes = [dp, vp, i32, vp, i32, fp, fp, fp, vp, sz, vp]
    lib.mvf_sgd_workspace_bytes.restype = sz
    lib.mvf_sgd_workspace_bytes.argtypes = [i64]
    lib.mvf_sgd_nesterov_step.restype = i32
    lib.mvf_sgd_nesterov_step.argtypes = [fp, fp, fp, i64, f32, f32, f32, f32, f32, i32, fp, vp, sz, vp]
    lib.mvf_sgd_step_segments.restype = i32
    lib.mvf_sgd_step_segments.argtypes = [fp, fp, fp, i64, f32, f32, f32, f32, f32, i32, i32, vp, i32, fp, vp, sz, vp]
    lib.mvf_plan_run.restype = i32
    lib.mvf_plan_run.argtypes = [C.POINTER(PlanOp), i32, C.POINTER(C.c_ulonglong), C.POINTER(C.c_float), C.POINTER(C.c_int)]
    # Callers pass device pointers as plain Python ints (train_engine._p): without declared argtypes ctypes would truncate them to c_int silently.  Every
    # function the header declares must therefore have its signature declared above.
    missing = [n for n in declared_symbols() if hasattr(lib, n) and getattr(lib, n).argtypes is None]
    if missing:
        raise ImportError("mvfnet_amd._lib: no argtypes declared for %s" % missing)
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = lib.mvf_last_error().decode("utf-8", "replace")
        raise RuntimeError("%s failed: %s (%s)" % (what or "libmvfnet_hip", ERRNAMES.get(rc, rc), msg))


def declared_symbols():
    """Every function name declared in include/mvfnet_hip.h (parsed; used by the symbol-export test)."""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "mvfnet_hip.h")
    txt = open(hdr).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b((?:mvf|mvfnet)_[a-z0-9_]+)\s*\(", txt)))


lib = _load()
