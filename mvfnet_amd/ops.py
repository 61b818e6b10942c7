"""torch.autograd wrappers over the C ABI (include/mvfnet_hip.h).  PyTorch here is plumbing only: device
memory (tensors), the current HIP stream, and autograd's tape.  All arithmetic of the hot path runs in
libmvfnet_hip.so; CPU tensors are rejected (no fallback)."""
import ctypes as C

import torch

from . import _lib
from ._lib import MvfDesc, check, lib

_DT = {torch.float32: _lib.MVF_F32, torch.bfloat16: _lib.MVF_BF16}


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_gpu(x, what):
    if not x.is_cuda:
        raise RuntimeError("%s: mvfnet_amd runs on MI355X (HIP) tensors only -- got a %s tensor. There is no CPU "
                           "fallback in the product path (the CPU restatement lives in oracle/ for tests)." % (what, x.device))
    if x.dtype not in _DT:
        raise TypeError("%s: dtype %s not supported (float32 / bfloat16)" % (what, x.dtype))


def _layout_of(x):
    if x.dim() != 4:
        raise ValueError("expected a (N*T, C, H, W) tensor, got shape %s" % (tuple(x.shape),))
    if x.is_contiguous():
        return _lib.MVF_NCHW
    if x.is_contiguous(memory_format=torch.channels_last):
        return _lib.MVF_NHWC
    return None


def _f32c(t):
    return None if t is None else t.detach().to(torch.float32).contiguous()


def _desc(x, layout, n_segment, cs, mode_bits):
    nt, c, h, w = x.shape
    return MvfDesc(nt, c, h, w, n_segment, cs, mode_bits, layout, _DT[x.dtype])


def _empty_like_layout(x, layout):
    return torch.empty_like(x, memory_format=torch.channels_last if layout == _lib.MVF_NHWC else torch.contiguous_format)


def _taps(w, cs):
    return _f32c(w).reshape(cs, 3)


class _MVFProper(torch.autograd.Function):
    """MVF.forward minus self.net (reference MVF.py:104-137) as one autograd node."""

    @staticmethod
    def forward(ctx, x, wt, wh, ww, gamma, beta, n_segment, cs, mode_bits, share, training, eps, momentum,
                running_mean, running_var):
        _require_gpu(x, "MVF")
        layout = _layout_of(x)
        use_hs_ = gamma is not None
        nhwc_train_ok = cs % 4 == 0 and x.shape[1] % 4 == 0        # mvf_fwd_train / mvf_bwd in MVF_NHWC need 4-channel groups
        if layout is None or (layout == _lib.MVF_NHWC and not nhwc_train_ok and
                              ((use_hs_ and training) or any(ctx.needs_input_grad[:6]))):
            # odd channel counts: training / backward of a channels-last tensor goes through the NCHW kernels on a
            # contiguous copy
            x = x.contiguous()
            layout = _lib.MVF_NCHW
        d = _desc(x, layout, n_segment, cs, mode_bits)
        wt_ = _taps(wt, cs)
        wh_ = wt_ if share else (_taps(wh, cs) if wh is not None else None)
        ww_ = wt_ if share else (_taps(ww, cs) if ww is not None else None)
        if not (mode_bits & _lib.VIEW_H):
            wh_ = None
        if not (mode_bits & _lib.VIEW_W):
            ww_ = None
        out = _empty_like_layout(x, layout)
        use_hs = gamma is not None
        need_grad = any(ctx.needs_input_grad[:6])
        mean = invstd = None
        g32, b32 = _f32c(gamma), _f32c(beta)
        if use_hs and training:
            for t in (running_mean, running_var):          # mvf_fwd_train updates them in place as fp32 (cs * 4 bytes each)
                if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
                    raise TypeError("MVF: BatchNorm running statistics must be contiguous float32 buffers in training mode (got %s); "
                                    "keep the module's bn in fp32 when casting the model" % t.dtype)
            ws = torch.empty(lib.mvf_fwd_train_workspace_bytes(C.byref(d)), dtype=torch.uint8, device=x.device)
            mean = torch.empty(cs, dtype=torch.float32, device=x.device)
            invstd = torch.empty(cs, dtype=torch.float32, device=x.device)
            check(lib.mvf_fwd_train(C.byref(d), _ptr(x), _ptr(out), _ptr(wt_), _ptr(wh_), _ptr(ww_), _ptr(g32), _ptr(b32),
                                    C.c_float(eps), C.c_float(momentum), _ptr(running_mean), _ptr(running_var),
                                    _ptr(mean), _ptr(invstd), _ptr(ws), ws.numel(), _stream()), "mvf_fwd_train")
        else:
            scale = shift = None
            if use_hs:
                mean = running_mean.to(torch.float32)
                invstd = 1.0 / torch.sqrt(running_var.to(torch.float32) + eps)
                scale = (g32 * invstd).contiguous()
                shift = (b32 - mean * scale).contiguous()
            check(lib.mvf_fwd_infer(C.byref(d), _ptr(x), _ptr(out), _ptr(wt_), _ptr(wh_), _ptr(ww_), _ptr(scale),
                                    _ptr(shift), _stream()), "mvf_fwd_infer")
        if need_grad:
            ctx.save_for_backward(x, wt_, wh_, ww_, g32, b32, mean, invstd)
            ctx.cfg = (layout, n_segment, cs, mode_bits, share, bool(use_hs and training),
                       wt.shape, None if wh is None else wh.shape, None if ww is None else ww.shape,
                       wt.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        x, wt_, wh_, ww_, g32, b32, mean, invstd = ctx.saved_tensors
        layout, n_segment, cs, mode_bits, share, training, s_t, s_h, s_w, wdtype = ctx.cfg
        g = g.contiguous(memory_format=torch.channels_last) if layout == _lib.MVF_NHWC else g.contiguous()
        d = _desc(x, layout, n_segment, cs, mode_bits)
        dx = _empty_like_layout(x, layout)
        dev = x.device
        dwt = torch.empty(cs, 3, dtype=torch.float32, device=dev)
        dwh = torch.empty(cs, 3, dtype=torch.float32, device=dev)
        dww = torch.empty(cs, 3, dtype=torch.float32, device=dev)
        use_hs = g32 is not None
        dgamma = torch.empty(cs, dtype=torch.float32, device=dev) if use_hs else None
        dbeta = torch.empty(cs, dtype=torch.float32, device=dev) if use_hs else None
        ws = torch.empty(lib.mvf_bwd_workspace_bytes(C.byref(d)), dtype=torch.uint8, device=dev)
        check(lib.mvf_bwd(C.byref(d), _ptr(g), _ptr(x), _ptr(wt_), _ptr(wh_), _ptr(ww_), _ptr(g32), _ptr(b32), _ptr(mean),
                          _ptr(invstd), int(training), _ptr(dx), _ptr(dwt), _ptr(dwh), _ptr(dww), _ptr(dgamma),
                          _ptr(dbeta), _ptr(ws), ws.numel(), _stream()), "mvf_bwd")
        if share:   # one weight serves all views (MVF.py:114-116): grads add up
            dwt = dwt + dwh + dww
            dwh = dww = None
        gt = dwt.reshape(s_t).to(wdtype)
        gh = dwh.reshape(s_h).to(wdtype) if (s_h is not None and dwh is not None and mode_bits & _lib.VIEW_H) else None
        gw = dww.reshape(s_w).to(wdtype) if (s_w is not None and dww is not None and mode_bits & _lib.VIEW_W) else None
        return (dx, gt, gh, gw, dgamma, dbeta) + (None,) * 9


def mvf_proper(x, wt, wh, ww, gamma, beta, n_segment, cs, mode="THW", share=False, training=False, eps=1e-5,
               momentum=0.1, running_mean=None, running_var=None):
    """Functional MVF-proper.  wt/wh/ww: Conv3d-shaped depthwise weights; gamma=None <=> use_hs=False."""
    return _MVFProper.apply(x, wt, wh, ww, gamma, beta, n_segment, cs, _lib.MODE_BITS[mode], share, training, eps,
                            momentum, running_mean, running_var)
