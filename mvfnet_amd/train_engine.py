"""Training engine: one MVFNet train step (forward with batch-statistics BN, loss, backward, clip + SGD-nesterov) as a
fixed sequence of HIP launches through the C ABI.

Host-side mirror of Recognizer2D.forward_train (reference codes/models/recognizers/recognizer2d.py:132-149),
Bottleneck.forward in train mode (codes/models/backbones/resnet.py:208-244, all BN layers on batch statistics:
norm_eval=False in the shipped configs), MVF.forward in train mode (codes/models/modules/MVF.py:104-138), the head +
loss (heads/tsn_clshead.py:71-98, heads/base.py:40-45), autograd's backward of all of it, and
DistOptimizerHook.after_train_iter (codes/core/dist_utils.py:61-67).

Layout: activations are channels-last matrices [m = n*h*w][c] in fp32.  Every parameter of the model lives in ONE
flat fp32 buffer (the nn.Parameters are re-pointed at views of it), gradients in a second flat buffer and momentum in
a third, so the data-parallel gradient exchange is a single in-place all-reduce of the flat gradient (what the
reference's `_allreduce_coalesced` builds by copying) and the optimizer is one fused kernel.

Kept per conv for the backward: its raw output z (pre-BN) and, for mid-block convs, a = relu(bn(z)); BN statistics.
Nothing is recomputed except ReLU / hard-swish masks (from z and the folded scale/shift).
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import ConvDesc, MvfDesc, check, lib

F32 = _lib.MVF_F32

# [r5] Descriptors are memoised by their field values (shapes are static from step to step): building a 19-field ctypes structure per launch and asking the library for
# mvf_conv2d_stats_rows each time was ~1.5 ms of host time per step (~300 + ~100 calls).
_ConvDescT, _MvfDescT = ConvDesc, MvfDesc
_DESC_CACHE = {}


def ConvDesc(*a):      # noqa: F811
    d = _DESC_CACHE.get(a)
    if d is None:
        d = _DESC_CACHE[a] = _ConvDescT(*a)
    return d


def MvfDesc(*a):       # noqa: F811
    k = ("mvf",) + a
    d = _DESC_CACHE.get(k)
    if d is None:
        d = _DESC_CACHE[k] = _MvfDescT(*a)
    return d


def _stats_rows(d):
    # kept ON the descriptor object (not in a table keyed by id(d): a descriptor built outside the memo and collected would hand its id -- and a stale row
    # count, i.e. an undersized partial-sum buffer -- to the next object allocated there)
    r = d.__dict__.get("_rows")
    if r is None:
        r = d._rows = lib.mvf_conv2d_stats_rows(C.byref(d))
    return r


def _p(t):
    # a plain int (every entry point declares its argtypes, so ctypes converts it): ~3500 calls per step, the c_void_p object each used to build was 0.5 ms of host time
    return t.data_ptr() if t is not None else None


from .policy import policy as _policy      # MVF_POLICY="name=value,...": the one environment variable of the switches below


_STREAM = [None]      # cached HIP stream handle of the stream the engine is launching on (torch.cuda.current_stream() costs ~8 us)
_MAINH = [None]       # handle of the launch stream while an engine entry point is running (_on_stream(main=True)); None outside: _conv_ws asks torch then


def _st():
    return _STREAM[0] if _STREAM[0] is not None else C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _on_stream(object):
    """with _on_stream(torch_stream_or_None): every _st() inside returns that stream's handle.  main=True (the engine's entry points): that stream also is the
    LAUNCH stream for the duration of the block -- _conv_ws() compares against it instead of asking torch -- and stops being it on exit, so a launch made outside
    an entry point, or by another engine on another stream, never inherits a stale handle."""

    def __init__(self, stream=None, main=False):
        self.h = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        self.main = main

    def __enter__(self):
        self.old, self.oldm = _STREAM[0], _MAINH[0]
        _STREAM[0] = self.h
        if self.main:
            _MAINH[0] = self.h.value

    def __exit__(self, *a):
        _STREAM[0] = self.old
        _MAINH[0] = self.oldm


_SIDE_WS = {}


def _conv_ws(device):
    """Stream-K scratch of the conv kernel: one buffer per stream the engine launches convs on (two convs running at the same time
    must not share the partial-tile slots and flags)."""
    from .engine import _sk_workspace
    h = _STREAM[0]
    if h is None or h.value == (_MAINH[0] if _MAINH[0] is not None else torch.cuda.current_stream().cuda_stream):
        return _sk_workspace(device)
    key = (str(device), h.value)
    ws = _SIDE_WS.get(key)
    if ws is None:
        ws = torch.empty(lib.mvf_conv2d_workspace_bytes(None), dtype=torch.uint8, device=device)
        _SIDE_WS[key] = ws
    return ws


class _BN(object):
    """Handles to one BatchNorm's parameters / buffers / gradient slots + per-step statistics."""

    def __init__(self, bn, eng, name):
        self.eng = eng
        self.c = bn.num_features
        self.mod = bn
        self.gamma, self.beta = bn.weight, bn.bias
        self.dgamma, self.dbeta = eng.grad_of(bn.weight), eng.grad_of(bn.bias)
        self.eps, self.momentum = bn.eps, (bn.momentum if bn.momentum is not None else 0.1)
        for t in (bn.running_mean, bn.running_var):       # the kernels read and update them in place as fp32
            if t is None or t.dtype != torch.float32 or not t.is_contiguous():
                raise TypeError("train engine: BatchNorm running statistics must be contiguous float32 buffers (got %s)" % (None if t is None else t.dtype))
        dev = bn.weight.device
        self.mean = torch.empty(self.c, device=dev)
        self.invstd = torch.empty(self.c, device=dev)
        self.scale = torch.empty(self.c, device=dev)
        self.shift = torch.empty(self.c, device=dev)

    @property
    def frozen(self):
        """The module is in eval mode (ResNet(norm_eval=True).train(), reference resnet.py:496-505): normalise with the running
        statistics, leave them untouched; gamma / beta still get gradients."""
        return not self.mod.training

    def use_running_stats(self):
        """Coefficients of a frozen-statistics BN for this step: mean / invstd from the running buffers, folded scale / shift."""
        check(lib.mvf_bn_fold(_p(self.gamma), _p(self.beta), _p(self.mod.running_mean), _p(self.mod.running_var), C.c_float(self.eps), self.c,
                              _p(self.scale), _p(self.shift), _st()), "mvf_bn_fold")
        self.mean.copy_(self.mod.running_mean)
        torch.rsqrt(self.mod.running_var + self.eps, out=self.invstd)
        if getattr(self, "_zero", None) is None:
            self._zero = torch.zeros(self.c, device=self.mean.device)

    def stats(self, z, m, eng):
        if self.frozen:
            return self.use_running_stats()
        ws = eng.workspace(lib.mvf_bn_workspace_bytes(m, self.c))
        check(lib.mvf_bn_train_stats(_p(z), m, self.c, _p(self.gamma), _p(self.beta), C.c_float(self.eps), C.c_float(self.momentum),
                                     _p(self.mod.running_mean), _p(self.mod.running_var), _p(self.mean), _p(self.invstd), _p(self.scale),
                                     _p(self.shift), _p(ws), ws.numel(), eng.dt, _st()), "mvf_bn_train_stats")
        self._count()

    def finalize(self, part, nblk, m):
        """Batch statistics from the per-tile partial sums the conv epilogue produced (same K = old running mean)."""
        check(lib.mvf_bn_train_finalize(_p(part), nblk, m, self.c, _p(self.gamma), _p(self.beta), C.c_float(self.eps), C.c_float(self.momentum),
                                        _p(self.mod.running_mean), _p(self.mod.running_var), _p(self.mean), _p(self.invstd), _p(self.scale),
                                        _p(self.shift), _st()), "mvf_bn_train_finalize")
        self._count()

    def _count(self):
        """num_batches_tracked += 1 -- TrainEngine keeps every BatchNorm's counter as a view of ONE int64 buffer and bumps them
        all with a single launch per step (62 tiny kernels, or worse 62 tiny copies under torch._foreach_add_, otherwise)."""
        if getattr(self.eng, "_nbt_flat", None) is None:
            self.mod.num_batches_tracked += 1
        else:
            self.eng._nbt_touched = True

    def apply(self, z, m, act, residual=None, rbn=None, bits=False):
        """out = act(bn(z) [+ residual]); bits=True also keeps the sign bits of out ([m][c/4] bytes) for the backward masks."""
        out = self.eng.buf((id(self), "apply"), z.shape, z.dtype)
        sb = self.eng.buf((id(self), "bits"), (m, self.c // 4), torch.uint8) if bits else None
        check(lib.mvf_bn_apply_bits(_p(z), m, self.c, _p(self.scale), _p(self.shift), _p(residual), _p(rbn.scale if rbn else None),
                                    _p(rbn.shift if rbn else None), act, _p(out), _p(sb), self.eng.dt, _st()), "mvf_bn_apply")
        return (out, sb) if bits else out

    def apply_colmeans(self, z, m, act, mean_out):
        """[r5] out = act(bn(z)) + the column means of what is stored into mean_out -- conv3's a_mean for the Gram form of bn3's statistics
        (_TConv.gram_stats) without a pass over a2."""
        out = self.eng.buf((id(self), "apply"), z.shape, z.dtype)
        ws = self.eng.workspace(max(lib.mvf_bn_workspace_bytes(m, self.c), 4608 * self.c * 8))        # (the apply plan aims at 4096 workgroups: one partial row each)
        check(lib.mvf_bn_apply_colmeans(_p(z), m, self.c, _p(self.scale), _p(self.shift), act, _p(out), _p(mean_out), _p(ws), ws.numel(), self.eng.dt, _st()),
              "mvf_bn_apply_colmeans")
        return out

    def backward(self, g, g_pitch, z, m, eng, mask_mode, ymask=None, gm_out=None, sums_done=False):
        """dgamma/dbeta into the flat grad buffer; returns dz.  sums_done: dgamma / dbeta were already produced by the data
        gradient that wrote g (_TConv.dgrad_bnsums)."""
        if not sums_done:
            self._reduce(g, g_pitch, z, m, eng, mask_mode, ymask, gm_out)
        return self._apply_bwd(g, g_pitch, z, m, eng, mask_mode, ymask, gm_out)

    @staticmethod
    def backward_pair(a, b, g, g_pitch, za, zb, m, eng, bits):
        """Backward of both BatchNorms of a downsample block, out = relu(a(za) + b(zb)): the shared g and sign bits are read once
        (mvf_bn_bwd_pair).  Returns (dza, dzb); results equal a.backward(...), b.backward(...) bit for bit."""
        nb = lib.mvf_bn_workspace_bytes(m, a.c)
        ws = eng.workspace(2 * nb)
        dza, dzb = eng.buf((id(a), "dz"), za.shape, za.dtype), eng.buf((id(b), "dz"), zb.shape, zb.dtype)
        check(lib.mvf_bn_bwd_pair(_p(g), g_pitch, _p(za), _p(zb), _p(bits), m, a.c, _p(a.gamma), _p(a.mean), _p(a.invstd), _p(a.dgamma), _p(a.dbeta),
                                  _p(b.gamma), _p(b.mean), _p(b.invstd), _p(b.dgamma), _p(b.dbeta), _p(dza), _p(dzb), _p(ws), ws.numel(), eng.dt, _st()),
              "mvf_bn_bwd_pair")
        return dza, dzb

    def backward_wgrad(self, g, g_pitch, z, m, eng, mask_mode, ymask, conv, x, x_pitch, sums_done=False):
        """[r4] backward() with the weight gradient of `conv` (the pointwise conv that produced z from x) taken INSIDE the apply pass
        (mvf_bn_bwd_apply_wgrad): dz is bit-identical, conv.dw arrives through a slab reduce queued on the side stream."""
        if not sums_done:
            self._reduce(g, g_pitch, z, m, eng, mask_mode, ymask, None)
        return self._apply_bwd_wgrad(g, g_pitch, z, m, eng, mask_mode, ymask, conv, x, x_pitch)

    def _apply_bwd_wgrad(self, g, g_pitch, z, m, eng, mask_mode, ymask, conv, x, x_pitch):
        """Exactly one fused launch (bench.py brackets this call with HIP events) + the slab reduce handed to the side stream."""
        dz = eng.buf((id(self), "dz"), z.shape, z.dtype)
        ns = lib.mvf_bn_bwd_wgrad_splits(m, self.c, conv.cin, 1, mask_mode)
        nb = lib.mvf_bn_bwd_wgrad_slab_bytes(m, self.c, conv.cin, 1, mask_mode)
        slabs = eng.buf((id(conv), "wslab"), (nb // 4,), torch.float32)
        sg, sb = (self._zero, self._zero) if self.frozen else (self.dgamma, self.dbeta)
        check(lib.mvf_bn_bwd_apply_wgrad(_p(g), g_pitch, _p(z), _p(ymask) if mask_mode == 4 else None, m, self.c, _p(self.gamma), _p(self.mean), _p(self.invstd),
                                         _p(self.scale), _p(self.shift), _p(sg), _p(sb), mask_mode, _p(dz), _p(x), x_pitch, conv.cin, _p(slabs), nb,
                                         eng.dt, _st()), "mvf_bn_bwd_apply_wgrad")
        conv.slab_reduce(slabs, ns, eng)
        return dz

    @staticmethod
    def backward_pair_wgrad(a, b, g, g_pitch, za, zb, m, eng, bits, conv_a, xa, xa_pitch, conv_b, xb, xb_pitch):
        """[r4] backward_pair with conv_a's (and, when xb is given, conv_b's) weight gradient inside the apply pass (mvf_bn_bwd_pair_wgrad)."""
        nbw = lib.mvf_bn_workspace_bytes(m, a.c)
        ws = eng.workspace(2 * nbw)
        dza, dzb = eng.buf((id(a), "dz"), za.shape, za.dtype), eng.buf((id(b), "dz"), zb.shape, zb.dtype)
        k = conv_a.cin
        ns = lib.mvf_bn_bwd_wgrad_splits(m, a.c, k, 2, 4)
        nb = lib.mvf_bn_bwd_wgrad_slab_bytes(m, a.c, k, 2, 4)
        sa = eng.buf((id(conv_a), "wslab"), (nb // 4,), torch.float32)
        sb = eng.buf((id(conv_b), "wslab"), (nb // 4,), torch.float32) if xb is not None else None
        check(lib.mvf_bn_bwd_pair_wgrad(_p(g), g_pitch, _p(za), _p(zb), _p(bits), m, a.c, _p(a.gamma), _p(a.mean), _p(a.invstd), _p(a.dgamma), _p(a.dbeta),
                                        _p(b.gamma), _p(b.mean), _p(b.invstd), _p(b.dgamma), _p(b.dbeta), _p(dza), _p(dzb), _p(xa), xa_pitch,
                                        _p(xb), xb_pitch, k, _p(sa), _p(sb), nb, _p(ws), ws.numel(), eng.dt, _st()), "mvf_bn_bwd_pair_wgrad")
        conv_a.slab_reduce(sa, ns, eng)
        if xb is not None:
            conv_b.slab_reduce(sb, ns, eng)
        return dza, dzb

    def _reduce(self, g, g_pitch, z, m, eng, mask_mode, ymask, gm_out):
        ws = eng.workspace(lib.mvf_bn_workspace_bytes(m, self.c))
        check(lib.mvf_bn_bwd_reduce(_p(g), g_pitch, _p(z), _p(ymask), m, self.c, _p(self.mean), _p(self.invstd), _p(self.scale),
                                    _p(self.shift), mask_mode, _p(gm_out), _p(self.dgamma), _p(self.dbeta), _p(ws), ws.numel(), eng.dt, _st()),
              "mvf_bn_bwd_reduce")

    def _apply_bwd(self, g, g_pitch, z, m, eng, mask_mode, ymask, gm_out):
        src, pitch, mode = (gm_out, self.c, 0) if gm_out is not None else (g, g_pitch, mask_mode)
        dz = eng.buf((id(self), "dz"), z.shape, z.dtype)
        # frozen statistics: mean and variance do not depend on z, so dz = gamma * invstd * (masked g) -- the batch-statistics
        # formula with its two correction sums set to zero (dgamma / dbeta themselves are still the real sums)
        sg, sb = (self._zero, self._zero) if self.frozen else (self.dgamma, self.dbeta)
        check(lib.mvf_bn_bwd_apply_masked(_p(src), pitch, _p(z), _p(ymask) if mode in (1, 4) else None, m, self.c, _p(self.gamma), _p(self.mean),
                                          _p(self.invstd), _p(self.scale), _p(self.shift), _p(sg), _p(sb), mode, _p(dz), eng.dt,
                                          _st()), "mvf_bn_bwd_apply")
        return dz


class _TConv(object):
    """One conv's parameter handle, per-step packed weights (forward and data-gradient) and launch helpers."""

    def __init__(self, conv, eng, stem=False):
        self.eng = eng
        self.w = conv.weight
        self.dw = eng.grad_of(conv.weight)
        self.cout, self.cin, self.kh, self.kw = conv.weight.shape
        self.stride, self.pad = conv.stride[0], conv.padding[0]
        self.stem = stem
        dev = conv.weight.device
        td = eng.tdtype
        if stem:
            self.wp = torch.empty(self.cout, self.kh, 8, 4, device=dev, dtype=td)
            self.wd = None
        else:
            # a 1x1 kernel's fp32 OIHW storage (cout, cin, 1, 1) already IS the packed [cout][1][1][cin] layout: no forward pack
            same = self.kh == 1 and self.kw == 1 and td == torch.float32
            self.wp = self.w if same else torch.empty(self.cout, self.kh, self.kw, self.cin, device=dev, dtype=td)
            self.wd = torch.empty(self.cin, self.kh, self.kw, self.cout, device=dev, dtype=td)

    def pack(self, need_dgrad=True):
        if self.stem:
            check(lib.mvf_pack_conv_weight(_p(self.w), self.cout, self.cin, self.kh, self.kw, 8, 4, None, _p(self.wp), self.eng.dt, _st()), "pack")
            return
        if self.wp is not self.w:
            check(lib.mvf_pack_conv_weight(_p(self.w), self.cout, self.cin, self.kh, self.kw, self.kw, self.cin, None, _p(self.wp), self.eng.dt, _st()), "pack")
        if need_dgrad:
            self.pack_dgrad()

    def pack_jobs(self):
        """(forward job | None, data-gradient job | None) as (w, out, cout, cin, kh, kw, kw_pad, cin_pad, kind) tuples for the
        batched pack kernel -- the same packs as pack() / pack_dgrad()."""
        if self.stem:
            return (self.w, self.wp, self.cout, self.cin, self.kh, self.kw, 8, 4, 0), None
        fwd = None if self.wp is self.w else (self.w, self.wp, self.cout, self.cin, self.kh, self.kw, self.kw, self.cin, 0)
        return fwd, (self.w, self.wd, self.cout, self.cin, self.kh, self.kw, self.kw, self.cin, 1)

    def pack_dgrad(self):
        if not self.stem:
            check(lib.mvf_pack_conv_weight_dgrad(_p(self.w), self.cout, self.cin, self.kh, self.kw, _p(self.wd), self.eng.dt, _st()), "pack_dgrad")

    def desc(self, n, h, w, ho, wo, x_pitch, split_c=0):
        if self.stem:
            return ConvDesc(n, h, w, 32, self.cout, self.kh, 1, 2, 0, ho, wo, 4, self.eng.dt, 0, 0, 0, 0)
        return ConvDesc(n, h, w, self.cin, self.cout, self.kh, self.kw, self.stride, self.pad, ho, wo, x_pitch, self.eng.dt, 0, split_c, split_c, 0)

    def out_hw(self, h, w):
        return (h + 2 * self.pad - self.kh) // self.stride + 1, (w + 2 * self.pad - self.kw) // self.stride + 1

    def forward(self, x, n, h, w, x_pitch=None, x2=None, split_c=0, ho=None, wo=None, bn=None, store=True):
        """z = conv(x); with `bn` the epilogue also accumulates that BatchNorm's batch statistics and they are finalised
        right away (no separate pass over z).  store=False ([r3], needs the fused statistics): a statistics-only pass, z is never written
        (returned as None) -- the block recomputes the conv wherever z would be read (forward_apply, bwd_sums, bwd_apply)."""
        if ho is None:
            ho, wo = self.out_hw(h, w)
        d = self.desc(n, h, w, ho, wo, x_pitch or self.cin, split_c)
        z = self.eng.buf((id(self), "z"), (n * ho * wo, self.cout)) if store else None
        ws = _conv_ws(x.device)
        if bn is None or not self.eng.fuse_stats or bn.frozen:
            self.launch_fwd(d, x, x2, z, ws, None, None)
            if bn is not None:
                bn.stats(z, n * ho * wo, self.eng)
            return z, ho, wo
        rows = _stats_rows(d)
        part = self.eng.buf((id(self), "part"), (rows, self.cout, 2), torch.float32)
        self.launch_fwd(d, x, x2, z, ws, part, bn.mod.running_mean)
        bn.finalize(part, rows, n * ho * wo)
        return z, ho, wo

    def forward_apply(self, x, n, h, w, bn, residual, rbn=None):
        """[r3] The block's last conv AGAIN, with bn's apply + residual + ReLU + sign bits as its epilogue (mvf_conv2d_nhwc_fwd_bnapply): the
        same `out` / bits as bn.apply(z3, residual, rbn, bits=True) reading the conv's narrow input instead of z3 -- bit for bit when the
        z3 that was stored came from the SAME kernel (same summation order: tests/test_conv_gpu.py).  In a z3-free block the statistics pass
        may be another kernel (pw_sums.hip: MFMA operands the other way round), so individual recomputed z3 elements can differ from the
        ones the statistics saw by one storage-type ulp: numerically benign, covered by test_z3_free_block_gradients_match_stored_z3_block."""
        ho, wo = self.out_hw(h, w)
        d = self.desc(n, h, w, ho, wo, self.cin)
        m = n * ho * wo
        out = self.eng.buf((id(bn), "apply"), (m, self.cout))
        bits = self.eng.buf((id(bn), "bits"), (m, self.cout // 4), torch.uint8)
        self.launch_fwd_apply(d, x, bn, residual, rbn, out, bits, _conv_ws(x.device))
        return out, bits

    def bwd_recompute(self, a_in, g, bits, n, h, w, bn):
        """[r3] BatchNorm backward of bn = the block's bn3 without a stored z3 (mvf_conv2d_nhwc_fwd_bnbwd_sums / _apply): dgamma / dbeta into
        the flat gradient buffer, returns dz3."""
        ho, wo = self.out_hw(h, w)
        d = self.desc(n, h, w, ho, wo, self.cin)
        m = n * ho * wo
        ws = _conv_ws(a_in.device)
        rows = _stats_rows(d)
        part = self.eng.buf((id(self), "bwpart"), (rows, self.cout, 2), torch.float32)
        self.launch_bwd_sums(d, a_in, g, bits, bn, part, ws)
        check(lib.mvf_bn_bwd_finalize(_p(part), rows, self.cout, _p(bn.dgamma), _p(bn.dbeta), _st()), "mvf_bn_bwd_finalize")
        dz = self.eng.buf((id(bn), "dz"), (m, self.cout))
        self.launch_bwd_apply(d, a_in, g, bits, bn, dz, ws)
        return dz

    def bwd_fused_ok(self, m):
        """[r4] The one-pass backward of a z3-free block's last conv is built for this shape (bf16, 64 -> 256 channels, pointwise, stride 1)."""
        return (self.eng.tdtype == torch.bfloat16 and self.kh == 1 and self.kw == 1 and self.stride == 1 and not self.stem and
                lib.mvf_conv1x1_bwd_fused_splits(m, self.cout, self.cin) > 0)

    def bwd_fused(self, a_in, g, bits, n, h, w, bn, bn_in=None, z_in=None, a_pitch=None, sums_done=False):
        """[r4] bwd_recompute + dgrad_bnsums + wgrad of a z3-free block's last conv with dz3 kept on chip (mvf_conv1x1_bwd_fused): bn's dgamma / dbeta
        from the sums pass as before, then ONE launch that returns the gradient of the conv input (bn_in's backward sums finalised) and leaves the
        weight gradient as partial slabs whose fixed-order reduce goes to the side stream.  bn_in = None: a conv that reads a block input (the downsample
        branch): the plain data gradient, no sums."""
        pitch = a_pitch or self.cin
        d = self.desc(n, h, w, h, w, pitch)
        m = n * h * w
        ws = _conv_ws(a_in.device)
        rows = _stats_rows(d)
        ns = lib.mvf_conv1x1_bwd_fused_splits(m, self.cout, self.cin)
        if not sums_done and self.eng.pair_ds_sums & 2:          # [r4] the one-branch form of csrc/pw_sums_pair.hip instead of pw_sums.hip
            part = self.eng.buf((id(self), "bwpart_pair"), (self.cout, 2 * ns, 2), torch.float32)
            self.launch_bwd_sums1(m, a_in, pitch, g, bits, bn, part, ns)
            check(lib.mvf_bn_bwd_finalize(_p(part), 2 * ns, self.cout, _p(bn.dgamma), _p(bn.dbeta), _st()), "mvf_bn_bwd_finalize")
        elif not sums_done:        # (a downsample block takes both branches' sums in one pass over g: _TBlock.backward)
            part = self.eng.buf((id(self), "bwpart"), (rows, self.cout, 2), torch.float32)
            self.launch_bwd_sums(d, a_in, g, bits, bn, part, ws)
            check(lib.mvf_bn_bwd_finalize(_p(part), rows, self.cout, _p(bn.dgamma), _p(bn.dbeta), _st()), "mvf_bn_bwd_finalize")
        dx = self.eng.buf((id(self), "dx"), (m, self.cin))
        spart = self.eng.buf((id(self), "bnsums_fused"), (self.cin, 2 * ns, 2), torch.float32) if bn_in is not None else None
        slabs = self.eng.buf((id(self), "wslab"), (ns * self.cout * self.cin,), torch.float32)
        self.launch_bwd_fused(m, a_in, g, bits, bn, bn_in, z_in, dx, spart, ns, slabs, pitch)
        if bn_in is not None:
            check(lib.mvf_bn_bwd_finalize(_p(spart), 2 * ns, self.cin, _p(bn_in.dgamma), _p(bn_in.dbeta), _st()), "bn bwd finalize")
        self.slab_reduce(slabs, ns, self.eng)
        return dx

    def launch_bwd_sums1(self, m, a_in, a_pitch, g, bits, bn, part, ns):
        """Exactly one launch (bench.py brackets this call with HIP events)."""
        check(lib.mvf_conv1x1_bnbwd_sums_pair(_p(a_in), a_pitch, _p(self.wp), None, 0, None, _p(g), self.cout, _p(bits), m, self.cout, self.cin, _p(bn.mean),
                                              _p(bn.invstd), None, None, _p(part), None, 2 * ns, self.eng.dt, _st()), "bn backward sums (one branch)")

    def launch_bwd_fused(self, m, a_in, g, bits, bn, bn_in, z_in, dx, spart, ns, slabs, a_pitch=None):
        """Exactly one launch (bench.py brackets this call with HIP events)."""
        i = bn_in
        check(lib.mvf_conv1x1_bwd_fused(_p(a_in), a_pitch or self.cin, _p(self.wp), _p(g), self.cout, _p(bits), m, self.cout, self.cin, _p(bn.gamma), _p(bn.mean),
                                        _p(bn.invstd), _p(bn.dgamma), _p(bn.dbeta), _p(z_in) if i is not None else None, _p(i.mean) if i else None,
                                        _p(i.invstd) if i else None, _p(i.scale) if i else None, _p(i.shift) if i else None,
                                        _p(dx), _p(spart), 2 * ns, _p(slabs), slabs.numel() * 4, self.eng.dt, _st()), "conv1x1 backward fused")

    def launch_bwd_sums(self, d, a_in, g, bits, bn, part, ws):
        check(lib.mvf_conv2d_nhwc_fwd_bnbwd_sums(C.byref(d), _p(a_in), None, _p(self.wp), _p(g), _p(bits), _p(bn.mean), _p(bn.invstd), _p(part),
                                                 _p(ws), ws.numel(), _st()), "conv + bn backward sums")

    def launch_bwd_apply(self, d, a_in, g, bits, bn, dz, ws):
        check(lib.mvf_conv2d_nhwc_fwd_bnbwd_apply(C.byref(d), _p(a_in), None, _p(self.wp), _p(g), _p(bits), _p(bn.gamma), _p(bn.mean), _p(bn.invstd),
                                                  _p(bn.dgamma), _p(bn.dbeta), _p(dz), _p(ws), ws.numel(), _st()), "conv + bn backward apply")

    def launch_fwd_apply(self, d, x, bn, residual, rbn, out, bits, ws):
        """Exactly one implicit-GEMM launch (bench.py brackets this call with HIP events)."""
        check(lib.mvf_conv2d_nhwc_fwd_bnapply(C.byref(d), _p(x), None, _p(self.wp), _p(bn.scale), _p(bn.shift), _p(residual),
                                              _p(rbn.scale if rbn else None), _p(rbn.shift if rbn else None), _p(out), _p(bits),
                                              _p(ws), ws.numel(), _st()), "conv fwd + bn apply")

    def launch_fwd(self, d, x, x2, z, ws, part, shift):
        """Exactly one implicit-GEMM launch (bench.py brackets this call with HIP events)."""
        if part is None:
            check(lib.mvf_conv2d_nhwc_fwd_ws(C.byref(d), _p(x), _p(x2), _p(self.wp), None, None, _p(z), _p(ws), ws.numel(), _st()), "conv fwd")
        else:
            check(lib.mvf_conv2d_nhwc_fwd_stats(C.byref(d), _p(x), _p(x2), _p(self.wp), _p(z), _p(part), _p(shift), _p(ws), ws.numel(), _st()),
                  "conv fwd+stats")

    def wgrad(self, dz, x, n, h, w, ho, wo, eng, x_pitch=None, x2=None, split_c=0, on_main=False):
        """Weight gradient; issued on the engine's side stream so it overlaps the data-gradient / BN chain (they are
        independent given dz) and fills the CUs the other chain's partial tile waves leave idle."""
        d = self.desc(n, h, w, ho, wo, x_pitch or self.cin, split_c)
        kwr, cinr, kwp, cinp = (self.kw, self.cin, 8, 4) if self.stem else (self.kw, self.cin, self.kw, self.cin)
        nbytes = lib.mvf_conv2d_wgrad_workspace_bytes(C.byref(d))
        side = None if on_main else eng.side_stream()
        if side is None:
            ws = eng.workspace(nbytes)
            check(lib.mvf_conv2d_nhwc_wgrad(C.byref(d), _p(dz), _p(x), _p(x2), kwr, cinr, kwp, cinp, _p(self.dw), _p(ws), ws.numel(), _st()), "conv wgrad")
            return
        ws = eng.workspace(nbytes, side=True)

        def launch():
            check(lib.mvf_conv2d_nhwc_wgrad(C.byref(d), _p(dz), _p(x), _p(x2), kwr, cinr, kwp, cinp, _p(self.dw), _p(ws), ws.numel(), _st()), "conv wgrad")
        eng.on_side(launch)
        # dz / x / x2 are persistent engine buffers (eng.buf) or tensors the caller keeps alive until join_side()

    def fuses_wgrad(self, eng, m, c, mask_mode, nbn=1):
        """[r4] This conv's weight gradient can ride in the BatchNorm-backward apply pass that forms its dz (bf16 storage, pointwise,
        stride 1, a c x cin accumulator that fits one workgroup: csrc/bnbwd_wgrad.hip)."""
        return (eng.tdtype == torch.bfloat16 and self.kh == 1 and self.kw == 1 and self.stride == 1 and not self.stem and c == self.cout and
                lib.mvf_bn_bwd_wgrad_splits(m, c, self.cin, nbn, mask_mode) > 0)

    def slab_reduce(self, slabs, ns, eng):
        """dw <- fixed-order sum of the fused pass's partial slabs: only feeds the optimizer, so it goes to the side stream."""
        def launch():
            check(lib.mvf_wgrad_slab_reduce(_p(slabs), ns, self.cout, self.cin, _p(self.dw), _st()), "wgrad slab reduce")
        if eng.side_stream() is None:
            launch()
        else:
            eng.on_side(launch)

    def dgrad_bnsums(self, dz, n, ho, wo, h, w, bn, z):
        """Data gradient that also accumulates the backward sums of `bn` (the ReLU(BN(z)) its output feeds) in its epilogue and
        finalises dgamma / dbeta: bn.backward(..., sums_done=True) then only needs the apply pass.  A strided conv's data gradient
        runs as its parity classes, each adding its own run of partial rows."""
        d = ConvDesc(n, ho, wo, self.cout, self.cin, self.kh, self.kw, 1, self.kh - 1 - self.pad, h, w, self.cout, self.eng.dt, 0, 0, 0,
                     self.stride if self.stride > 1 else 0, 0)
        dx = self.eng.buf((id(self), "dx"), (n * h * w, self.cin))
        ws = _conv_ws(dz.device)
        rows = _stats_rows(d)
        part = self.eng.buf((id(self), "bnsums"), (rows, self.cin, 2), torch.float32)
        self.launch_dgrad_bnsums(d, dz, dx, z, bn, part, ws)
        check(lib.mvf_bn_bwd_finalize(_p(part), rows, self.cin, _p(bn.dgamma), _p(bn.dbeta), _st()), "bn bwd finalize")
        return dx

    def launch_dgrad_bnsums(self, d, dz, dx, z, bn, part, ws):
        """Exactly one implicit-GEMM launch (bench.py brackets this call with HIP events)."""
        check(lib.mvf_conv2d_nhwc_dgrad_bnsums(C.byref(d), _p(dz), _p(self.wd), _p(dx), _p(z), _p(bn.mean), _p(bn.invstd), _p(bn.scale), _p(bn.shift),
                                               _p(part), _p(ws), ws.numel(), _st()), "conv dgrad+bn sums")

    def dgrad(self, dz, n, ho, wo, h, w, residual=None, res_c0=0, res_bits=None, out_gate=None, gsum=None):
        """dx (n*h*w, cin) from dz (n*ho*wo, cout): a conv of dz with the flipped/transposed weights (+ residual, on output
        channels >= res_c0, gated per element by the sign bits res_bits when given).  [r5] out_gate: the sign bits of the tensor dx is the
        gradient of -- channels >= res_c0 of dx are gated by them, so the block below receives gm = g * [out > 0] as a tensor; gsum (dict(bn)): and the
        column sums of what is stored."""
        d = ConvDesc(n, ho, wo, self.cout, self.cin, self.kh, self.kw, 1, self.kh - 1 - self.pad, h, w, self.cout, self.eng.dt, 0, 0, 0,
                     self.stride if self.stride > 1 else 0, res_c0)
        dx = self.eng.buf((id(self), "dx"), (n * h * w, self.cin))
        ws = _conv_ws(dz.device)
        if out_gate is not None and gsum is not None:
            # [r5] ... + the column sums of the gated gradient on channels >= res_c0: the block below completes bn3's backward sums from its weight-gradient
            # GEMM (dzfree_q_sums), no pass over (gm, z3)
            bn = gsum["bn"]
            rows = _stats_rows(d)
            part = self.eng.buf((id(bn), "gsum_conv"), (self.cin, rows, 2), torch.float32)
            check(lib.mvf_conv2d_nhwc_fwd_resmask_gate_colsums(C.byref(d), _p(dz), None, _p(self.wd), _p(residual), _p(res_bits), _p(out_gate), _p(dx), _p(part), _p(ws),
                                                               ws.numel(), _st()), "conv dgrad (gated output + column sums)")
            bn._s1 = [None, 0, res_c0, part, rows]
        elif out_gate is not None:
            check(lib.mvf_conv2d_nhwc_fwd_resmask_gate(C.byref(d), _p(dz), None, _p(self.wd), None, _p(residual), _p(res_bits), _p(out_gate), _p(dx), _p(ws),
                                                       ws.numel(), _st()), "conv dgrad (gated output)")
        else:
            check(lib.mvf_conv2d_nhwc_fwd_resmask(C.byref(d), _p(dz), None, _p(self.wd), None, _p(residual), _p(res_bits), _p(dx), _p(ws), ws.numel(),
                                                  _st()), "conv dgrad")
        return dx

    # ---- [r5] the block's last conv + bn3 backward WITHOUT the dz3 tensor (csrc/bn_dzfree.hip) -------------------------------------------------
    def dzfree_ok(self, m):
        """Pointwise stride-1 conv, bf16 storage, shapes the split-operand data gradient and the prep kernel take."""
        return (self.eng.tdtype == torch.bfloat16 and self.kh == 1 and self.kw == 1 and self.stride == 1 and not self.stem and
                self.cout % 256 == 0 and self.cin % 64 == 0 and self.cout <= 4096 and m * (self.cout + self.cin) * 2 < 0x7ffffff0)

    def dzfree_dgrad(self, gm, a_in, bn, n, h, w, bn_in, z_in, frozen_zero=None):
        """da = dz W with dz = bn's backward of gm -- taken as ONE GEMM over [gm | a_in] with the weights [a.W ; -G] and a bias (mvf_bn_bwd_dzfree_prep),
        + bn_in's backward sums in the epilogue.  bn.dgamma / bn.dbeta must hold the sums already."""
        eng = self.eng
        c, k, m = self.cout, self.cin, n * h * w
        bd = eng.buf((id(self), "dzfree_w"), (k, c + k))
        bias = eng.buf((id(self), "dzfree_b"), (k,), torch.float32)
        sg, sb = (frozen_zero, frozen_zero) if frozen_zero is not None else (bn.dgamma, bn.dbeta)
        check(lib.mvf_bn_bwd_dzfree_prep(_p(self.wd), c, k, _p(bn.gamma), _p(bn.mean), _p(bn.invstd), _p(sg), _p(sb), m, _p(bd), _p(bias), eng.dt, _st()),
              "bn backward without dz: operands")
        d = ConvDesc(n, h, w, c + k, k, 1, 1, 1, 0, h, w, k, eng.dt, 0, c, c, 0, 0, c)        # x = a_in (pitch k, columns c .. c + k), x2 = gm (pitch c)
        dx = eng.buf((id(self), "dx"), (m, k))
        ws = _conv_ws(gm.device)
        rows = _stats_rows(d)
        part = eng.buf((id(self), "bnsums"), (rows, k, 2), torch.float32)
        self.launch_dzfree_dgrad(d, a_in, gm, bd, bias, dx, z_in, bn_in, part, ws)
        check(lib.mvf_bn_bwd_finalize(_p(part), rows, k, _p(bn_in.dgamma), _p(bn_in.dbeta), _st()), "bn bwd finalize")
        return dx

    def launch_dzfree_dgrad(self, d, a_in, gm, bd, bias, dx, z_in, bn_in, part, ws):
        """Exactly one implicit-GEMM launch (bench.py brackets this call with HIP events)."""
        check(lib.mvf_conv2d_nhwc_dgrad_bnsums_split(C.byref(d), _p(a_in), _p(gm), _p(bd), _p(bias), _p(dx), _p(z_in), _p(bn_in.mean), _p(bn_in.invstd),
                                                     _p(bn_in.scale), _p(bn_in.shift), _p(part), _p(ws), ws.numel(), _st()), "conv dgrad on [gm | a] + bn sums")

    def dzfree_q_sums(self, gm, a_in, bn, n, h, w, eng):
        """[r5] bn's dgamma / dbeta with NO pass over (gm, z3): Q = gm^T a_in (the weight-gradient GEMM of dzfree_wgrad, taken first and on the LAUNCH stream,
        sized for the whole chip) gives sum gm z3 = sum_k W Q, and the kernels that stored gm left its column sums (bn._s1: the block above's conv1 data
        gradient for the channels >= its MVF slice, its transposed stencil for the slice)."""
        c, k = self.cout, self.cin
        d = self.desc(n, h, w, h, w, k, 0)
        ws = eng.workspace(lib.mvf_conv2d_wgrad_workspace_bytes(C.byref(d)))
        self.launch_q(d, gm, a_in, ws, eng.dzfree_q_wgs)
        lo, rows_lo, c_split, hi, rows_hi = bn._s1
        assert c_split == 0 or lo is not None, "the MVF slice's column sums are missing"
        check(lib.mvf_bn_bwd_dzfree_sums(_p(self.dw), _p(self.wp), c, k, _p(bn.mean), _p(bn.invstd), _p(lo), rows_lo, c_split, _p(hi), rows_hi, _p(bn.dgamma),
                                         _p(bn.dbeta), eng.dt, _st()), "bn backward sums from the weight-gradient GEMM")
        bn._s1 = None

    def launch_q(self, d, gm, a_in, ws, wgs):
        """One weight-gradient GEMM + its slab reduce on the launch stream (bench.py brackets this call with HIP events)."""
        check(lib.mvf_conv2d_nhwc_wgrad_wgs(C.byref(d), _p(gm), _p(a_in), None, self.kw, self.cin, self.kw, self.cin, _p(self.dw), _p(ws), ws.numel(), wgs, _st()),
              "conv wgrad (launch stream)")

    def gram_ok(self):
        """The shapes mvf_bn_train_stats_gram and its Gram GEMM take: a pointwise stride-1 conv in bf16 storage, channel counts in whole MFMA tiles."""
        return (self.eng.tdtype == torch.bfloat16 and self.kh == 1 and self.kw == 1 and self.stride == 1 and not self.stem and
                self.cin % 32 == 0 and self.cout % 32 == 0)

    def gram_stats(self, a_in, bn, n, h, w, eng, means_done=False):
        """[r5] bn's batch statistics of z = conv(a_in) WITHOUT the conv (csrc/bn_dzfree.hip: mvf_bn_train_stats_gram): the Gram matrix of a_in (one
        k x k GEMM over the narrow tensor), its column means, and a c x k x k quadratic form.  gram / a_mean stay in their per-conv buffers: the dz3-free
        backward's weight-gradient correction needs exactly these (dzfree_wgrad(gram_done=True))."""
        c, k, m = self.cout, self.cin, n * h * w
        gram = eng.buf((id(self), "gram"), (k, k), torch.float32)
        amean = eng.buf((id(self), "amean"), (4, k), torch.float32)            # rows: mean, and three outputs of the statistics call nobody reads
        d = ConvDesc(n, h, w, k, k, 1, 1, 1, 0, h, w, k, eng.dt, 0, 0, 0, 0)
        ws = eng.workspace(max(lib.mvf_conv2d_wgrad_workspace_bytes(C.byref(d)), lib.mvf_bn_workspace_bytes(m, k), (eng.gram_stats_wgs + 8) * k * k * 4))
        if k not in eng._ones:
            eng._ones[k] = (torch.ones(k, device=a_in.device), torch.zeros(k, device=a_in.device))
        one, zero = eng._ones[k]
        self.launch_gram(d, a_in, gram, ws, eng.gram_stats_wgs)
        if not means_done:     # (means_done: the kernel that wrote a_in left its column means in amean[0] -- _BN.apply_colmeans)
            check(lib.mvf_bn_train_stats(_p(a_in), m, k, _p(one), _p(zero), C.c_float(1e-5), C.c_float(0.1), None, None, _p(amean[0]), _p(amean[1]),
                                         _p(amean[2]), _p(amean[3]), _p(ws), ws.numel(), eng.dt, _st()), "column means")
        check(lib.mvf_bn_train_stats_gram(_p(gram), _p(amean[0]), _p(self.wp), m, c, k, _p(bn.gamma), _p(bn.beta), C.c_float(bn.eps), C.c_float(bn.momentum),
                                          _p(bn.mod.running_mean), _p(bn.mod.running_var), _p(bn.mean), _p(bn.invstd), _p(bn.scale), _p(bn.shift), eng.dt,
                                          _st()), "bn statistics from the Gram matrix")
        bn._count()

    def launch_gram(self, d, a_in, gram, ws, wgs):
        """One k x k GEMM a^T a (+ its slab reduce) on the launch stream (bench.py brackets this call with HIP events)."""
        k = self.cin
        check(lib.mvf_conv2d_nhwc_wgrad_wgs(C.byref(d), _p(a_in), _p(a_in), None, 1, k, 1, k, _p(gram), _p(ws), ws.numel(), wgs, _st()), "gram (launch stream)")

    def dzfree_wgrad(self, gm, a_in, bn, n, h, w, eng, frozen_zero=None, q_done=False, gram_done=False):
        """dW = dz^T a_in without dz: Q = gm^T a_in (the usual weight-gradient GEMM, into dw), A2 = a_in^T a_in and the column means of a_in, then the
        correction kernel -- all on the side stream (they only feed the optimizer)."""
        c, k, m = self.cout, self.cin, n * h * w
        if not q_done:         # (dzfree_q_sums already left Q in dw)
            self.wgrad(gm, a_in, n, h, w, h, w, eng)
        gram = eng.buf((id(self), "gram"), (k, k), torch.float32)
        amean = eng.buf((id(self), "amean"), (4, k), torch.float32)            # rows: mean, and three outputs of the statistics call nobody reads
        d = ConvDesc(n, h, w, k, k, 1, 1, 1, 0, h, w, k, eng.dt, 0, 0, 0, 0)
        nbytes = max(lib.mvf_conv2d_wgrad_workspace_bytes(C.byref(d)), lib.mvf_bn_workspace_bytes(m, k))
        if k not in eng._ones:
            eng._ones[k] = (torch.ones(k, device=gm.device), torch.zeros(k, device=gm.device))
        one, zero = eng._ones[k]
        sg, sb = (frozen_zero, frozen_zero) if frozen_zero is not None else (bn.dgamma, bn.dbeta)
        side = eng.side_stream()
        ws = eng.workspace(nbytes, side=side is not None)

        def launch():
            if not gram_done:      # (the forward's statistics already left both in place: gram_stats)
                check(lib.mvf_conv2d_nhwc_wgrad(C.byref(d), _p(a_in), _p(a_in), None, 1, k, 1, k, _p(gram), _p(ws), ws.numel(), _st()), "gram")
                check(lib.mvf_bn_train_stats(_p(a_in), m, k, _p(one), _p(zero), C.c_float(1e-5), C.c_float(0.1), None, None, _p(amean[0]), _p(amean[1]),
                                             _p(amean[2]), _p(amean[3]), _p(ws), ws.numel(), eng.dt, _st()), "column means")
            check(lib.mvf_bn_bwd_dzfree_wgrad(_p(self.dw), _p(self.wp), _p(gram), _p(amean[0]), _p(bn.gamma), _p(bn.mean), _p(bn.invstd), _p(sg), _p(sb),
                                              m, c, k, eng.dt, _st()), "weight gradient without dz: correction")
        if side is None:
            launch()
        else:
            eng.on_side(launch)


class _TMvf(object):
    def __init__(self, mvf, eng):
        self.eng = eng
        self.cs, self.T = mvf.num_shift_channel, mvf.n_segment
        self.mode = _lib.MODE_BITS[mvf.mode]
        self.share, self.use_hs = mvf.share, mvf.use_hs
        cs = self.cs
        self.wt = mvf.shift_conv.weight
        self.wh = mvf.shift_conv.weight if mvf.share else (mvf.h_conv.weight if self.mode & 2 else None)
        self.ww = mvf.shift_conv.weight if mvf.share else (mvf.w_conv.weight if self.mode & 4 else None)
        self.dwt = eng.grad_of(mvf.shift_conv.weight)
        self.dwh = None if mvf.share or not (self.mode & 2) else eng.grad_of(mvf.h_conv.weight)
        self.dww = None if mvf.share or not (self.mode & 4) else eng.grad_of(mvf.w_conv.weight)
        self.bn = _BN(mvf.bn, eng, "mvf.bn") if self.use_hs else None
        dev = mvf.shift_conv.weight.device
        self.tmp_h = torch.empty(cs, 3, device=dev) if self.dwh is None else None     # scratch for views without own weights
        self.tmp_w = torch.empty(cs, 3, device=dev) if self.dww is None else None

    def desc(self, nt, h, w, c):
        return MvfDesc(nt, c, h, w, self.T, self.cs, self.mode, _lib.MVF_NHWC, self.eng.dt)

    def forward(self, x, nt, h, w, c, eng):
        m = nt * h * w
        d = self.desc(nt, h, w, c)
        y = self.eng.buf((id(self), "y"), (m, self.cs))
        if self.use_hs and eng.fuse_mvf_stats and eng.fuse_stats and not self.bn.frozen and self.cs % 4 == 0:
            # [r5] the stencil accumulates the BatchNorm's batch statistics over what it stores: no separate pass over y
            rows = lib.mvf_nhwc_stencil_stats_rows(C.byref(d), c, self.cs)
            part = self.eng.buf((id(self), "part"), (self.cs, rows, 2), torch.float32)
            self.launch_stencil_stats(d, x, c, y, part)
            self.bn.finalize(part, rows, m)
            return y, self.bn.apply(y, m, 2)
        self.launch_stencil(d, x, c, y, self.cs, 0, None, 0, None)
        if not self.use_hs:
            return y, y
        self.bn.stats(y, m, eng)
        return y, self.bn.apply(y, m, 2)

    def launch_stencil_stats(self, d, x, c, y, part):
        """Exactly one MVF stencil launch (bench.py brackets this call with HIP events)."""
        check(lib.mvf_nhwc_stencil_stats(C.byref(d), _p(x), c, _p(y), self.cs, _p(self.wt), _p(self.wh), _p(self.ww), _p(part), _p(self.bn.mod.running_mean),
                                         _st()), "mvf stencil + statistics")

    def backward(self, dxp, x, y, nt, h, w, c, eng, addend=None, addend_bits=None, out_gate=None, gsum=None):
        """dxp (m, c): grad w.r.t. the conv input [o | x_rest]; on return its first cs channels hold the grad w.r.t. x's slice
        (+ addend[:, :cs], the skip-connection gradient, when given: the conv epilogue added it to channels >= cs only)."""
        m = nt * h * w
        d = self.desc(nt, h, w, c)
        if self.use_hs:
            dy = self.bn.backward(dxp, c, y, m, eng, 3)
        else:
            dy = dxp[:, : self.cs].contiguous()
        nbytes = lib.mvf_nhwc_tapgrad_workspace_bytes(C.byref(d))
        dwh = self.dwh if self.dwh is not None else self.tmp_h
        dww = self.dww if self.dww is not None else self.tmp_w
        side = eng.side_stream() if (self.use_hs and not self.share) else None
        if side is None:
            ws = eng.workspace(nbytes)
            check(lib.mvf_nhwc_tapgrad(C.byref(d), _p(x), c, _p(dy), self.cs, _p(self.dwt), _p(dwh), _p(dww), _p(ws), ws.numel(), _st()), "mvf tapgrad")
            if self.share:   # one weight tensor serves every view (MVF.py:114-116): gradients add up
                self.dwt.view(self.cs, 3).add_(dwh.view(self.cs, 3) if self.mode & 2 else 0).add_(dww.view(self.cs, 3) if self.mode & 4 else 0)
        else:
            # the tap gradients only feed the optimizer: off the critical path, on the weight-gradient stream (x and dy are
            # persistent engine buffers, untouched until join_side())
            ws = eng.workspace(nbytes, side=True)

            def launch():
                check(lib.mvf_nhwc_tapgrad(C.byref(d), _p(x), c, _p(dy), self.cs, _p(self.dwt), _p(dwh), _p(dww), _p(ws), ws.numel(), _st()), "mvf tapgrad")
            eng.on_side(launch)
        self.launch_stencil(d, dy, self.cs, dxp, c, 1, addend, c if addend is not None else 0, addend_bits, out_gate, gsum)

    def launch_stencil(self, d, src, src_c, dst, dst_c, flip, addend, addend_c, addend_bits, out_gate=None, gsum=None):
        """Exactly one MVF stencil launch (plain: y = taps * x-slice; flip: the transposed stencil of the backward, + gated addend
        [, the result gated by out_gate: the block below then receives the slice of gm = g * [out > 0]]); bench.py brackets this call with HIP events."""
        if out_gate is not None and gsum is not None:        # [r5] ... + the slice's column sums (the block below's dzfree_q_sums)
            bn = gsum["bn"]
            rows = lib.mvf_nhwc_stencil_stats_rows(C.byref(d), src_c, dst_c)
            part = self.eng.buf((id(bn), "gsum_mvf"), (self.cs, rows, 2), torch.float32)
            check(lib.mvf_nhwc_stencil_gate_colsums(C.byref(d), _p(src), src_c, _p(dst), dst_c, _p(self.wt), _p(self.wh), _p(self.ww), flip, _p(addend), addend_c,
                                                    _p(addend_bits), _p(out_gate), _p(part), _st()), "mvf stencil (gated output + column sums)")
            bn._s1[0], bn._s1[1] = part, rows
            return
        if out_gate is not None:
            check(lib.mvf_nhwc_stencil_gate(C.byref(d), _p(src), src_c, _p(dst), dst_c, _p(self.wt), _p(self.wh), _p(self.ww), None, None, flip,
                                            _p(addend), addend_c, _p(addend_bits), _p(out_gate), _st()), "mvf stencil (gated output)")
            return
        check(lib.mvf_nhwc_stencil(C.byref(d), _p(src), src_c, _p(dst), dst_c, _p(self.wt), _p(self.wh), _p(self.ww), None, None, flip,
                                   _p(addend), addend_c, _p(addend_bits), _st()), "mvf stencil")


class _TBlock(object):
    def __init__(self, blk, eng):
        from .modules.MVF import MVF
        c1 = blk.conv1
        self.mvf = None
        if isinstance(c1, MVF):
            if c1.num_shift_channel:
                self.mvf = _TMvf(c1, eng)
            c1 = c1.net
        self.c1, self.b1 = _TConv(c1, eng), _BN(blk.bn1, eng, "bn1")
        self.c2, self.b2 = _TConv(blk.conv2, eng), _BN(blk.bn2, eng, "bn2")
        self.c3, self.b3 = _TConv(blk.conv3, eng), _BN(blk.bn3, eng, "bn3")
        self.cd = self.bd = None
        if blk.downsample is not None:
            self.cd, self.bd = _TConv(blk.downsample[0], eng), _BN(blk.downsample[1], eng, "down")
        self.split_ok = self.mvf is not None and self.mvf.cs % (32 if eng.tdtype == torch.float32 else 64) == 0

    def convs(self):
        return [c for c in (self.c1, self.c2, self.c3, self.cd) if c is not None]

    def fuse_apply(self, eng):
        """bn3's apply + residual + ReLU as the epilogue of a SECOND conv3 pass instead of a pass over z3 (eng.fuse_bn3_apply: 0 never,
        1 where it measured faster in the step -- planes <= 128 (layer1: 187 vs 260 us per block, layer2: 110 vs 126): conv3 reads a quarter
        of z3's bytes and the matrix cores idle; from planes = 256 on the extra GEMM costs more than the bytes save (layer3: 71 vs 64 us) --,
        2 every block)."""
        # ([r5] planes <= 256 -- with the Gram statistics (gram_fwd) layer3 would also drop its first conv3 pass -- measured WORSE, three alternations on one
        # box: C3 17.57-17.65 -> 17.99-18.09 ms, C4 30.07-30.16 -> 31.94-32.01: the apply epilogue on the 1024-wide GEMM costs more than the byte-bound pass)
        f = eng.fuse_bn3_apply
        return f == 2 or (f == 1 and self.c3.cin <= 128)

    def z3_free(self, eng, m2=None):
        """[r3] The block never stores z3 (reference resnet.py:229-244: out = relu(bn3(conv3(a2)) + identity)): a statistics-only conv3 pass, the
        fused apply pass, and bn3's backward on the recomputed conv (eng.z3_free).  Plain blocks only: a downsample block's paired backward
        reads z3 and z_d in one pass over g; needs batch statistics fused into the conv epilogue."""
        m2 = m2 or (1 << 16)           # [r5] the block's real pixel count when the caller knows it: the kernels' plan functions refuse m * c * 2 >= 2^31
        if self.cd is not None:
            return self.z3_free_ds(eng, m2)
        if not (eng.z3_free and self.fuse_apply(eng) and eng.fuse_stats and not self.b3.frozen):
            return False
        if self.q_z3_free(eng, m2):
            return True            # [r5] its backward reads no z3 at all (dzfree_q): the first conv3 pass only takes the statistics
        if (eng.fuse_bnwg & 8) and eng.tdtype == torch.bfloat16 and lib.mvf_bn_bwd_wgrad_splits(m2, self.c3.cout, self.c3.cin, 1, 4) > 0:
            return False           # A/B: stored z3 + conv3's weight gradient in bn3's backward apply instead
        # measured per block in the bf16 step (us; stored-z3 path -> recompute path): statistics-only pass 148 -> 79 (layer1) / 75 -> 52 (layer2),
        # backward sums 168 -> 205 / 102 -> 119 (the conv kernel's sum epilogue streams g at 2.7 TB/s, the BatchNorm kernel at 5.2), backward apply
        # 242 -> 204 / 116 -> 119: -70 per layer1 block, ~0 per layer2 block -> planes <= 64 by default, 2 = every block with the fused apply
        return eng.z3_free == 2 or self.c3.cin <= 64

    def z3_free_ds(self, eng, m2=1 << 16):
        """[r4] A DOWNSAMPLE block (reference resnet.py:227-233: out = relu(bn3(conv3(a2)) + bn_d(conv_d(x)))) that does not store z3 and reads neither z3 nor
        z_d in backward: forward = conv3's statistics-only pass + the fused apply pass with the stored z_d as the residual operand; backward = per branch
        the sums pass on the recomputed conv (pw_sums) and the one-pass kernel of csrc/pw_bwd_fused.hip (conv3: + bn2's sums; the downsample conv: the
        plain data gradient) instead of the paired BatchNorm backward (two dz tensors written, each read twice).  Where that kernel is built: both convs
        64 -> 256 channels pointwise stride 1 (layer1.0), bf16."""
        if not (eng.z3_free_ds and eng.z3_free and eng.fuse_c3_bwd and eng.fuse_bn_bwd_sums and eng.fuse_stats and self.fuse_apply(eng) and eng.tdtype == torch.bfloat16):
            return False
        if self.b3.frozen or self.bd.frozen or self.cd.stride != 1 or self.cd.kh != 1 or self.cd.kw != 1 or self.cd.cin != self.c3.cin or self.cd.cout != self.c3.cout:
            return False
        # [r5] asked with the block's REAL pixel count: the backward of this form has no fallback, so the forward must not choose it where the one-pass
        # kernel (stride 1: the downsample branch sees the same m2 pixels) refuses the shape
        return lib.mvf_conv1x1_bwd_fused_splits(m2, self.c3.cout, self.c3.cin) > 0

    above = None       # the block whose backward produces this block's output gradient (set by the engine / BlockTrainer over a chain)

    def q_policy(self, eng, m2):
        """eng.dzfree_q: 0 off, 1 planes <= dzfree_q_maxk (the GEMM costs 2 m c k flops against the sums pass's 4 m c bytes: half the pass at k = 128, even at
        256, twice at 512) and only where that pass is long enough to pay for the GEMM's extra launches on the launch stream (>= 160 MB of (gm, z3)), 2 every block."""
        return eng.dzfree_q == 2 or (eng.dzfree_q == 1 and eng.dzfree_q_mink <= self.c3.cin <= eng.dzfree_q_maxk and m2 * self.c3.cout * 4 >= 160e6)

    def _dzfree_base(self, eng, m2):
        if not eng.dzfree or self.cd is not None or not eng.fuse_bn_bwd_sums or not self.c3.dzfree_ok(m2 or (1 << 16)):
            return False
        return eng.dzfree == 2 or self.c3.cin >= 256

    def gram_fwd(self, eng, m2):
        """[r5] bn3's batch statistics from the Gram matrix of a2 instead of a conv3 pass (eng.gram_stats): blocks that apply bn3 in a second conv3 pass
        (fuse_apply) and never read z3 in backward -- the z3-free blocks of layer1, and the dz3-free blocks whose sums come from Q (q_z3_free)."""
        if not (eng.gram_stats and eng.tdtype == torch.bfloat16 and self.cd is None and self.c3.gram_ok() and self.fuse_apply(eng) and eng.fuse_stats and
                not self.b3.frozen and eng.z3_free):
            return False
        return self.z3_free(eng, m2)

    def q_z3_free(self, eng, m2):
        """[r5] A dz3-free block whose bn3 sums come from the producers' column sums + Q (dzfree_q) reads z3 nowhere in backward, so where bn3's apply
        already is a second conv3 pass (fuse_apply) and its statistics come from the Gram matrix of a2 (gram_stats) there is no first conv3 pass and no
        stored z3.  Decided at forward time; should the block above not deliver the column sums after all, backward falls back to bn3's backward on the
        recomputed conv."""
        return bool(eng.gram_stats and self.c3.gram_ok() and eng.gate_producer and self.above is not None and
                    not self.b3.frozen and self.fuse_apply(eng) and self._dzfree_base(eng, m2) and self.q_policy(eng, m2))

    def dzfree(self, eng, m2=None):
        """[r5] bn3's backward without the dz3 tensor (csrc/bn_dzfree.hip): plain blocks (no downsample branch) with a STORED z3 whose last conv
        is pointwise; eng.dzfree: 0 off, 1 planes >= 256 (where dz3 is the widest tensor and no fused pass exists), 2 every eligible block."""
        if self.cd is not None or (self.z3_free(eng, m2) and not self.q_z3_free(eng, m2 or (1 << 16))):
            return False
        return self._dzfree_base(eng, m2)

    def wants_gated_gradient(self, eng):
        """The sign bits of this block's output when its backward wants gm = g * [out > 0] as a tensor (the dz3-free path reads it by LDS-DMA):
        the block ABOVE then gates the gradient it produces (conv1's data-gradient epilogue, the MVF transposed stencil)."""
        s = self.saved
        if s is None or not eng.gate_producer or not self.dzfree(eng, s["out"].shape[0]):
            return None
        # [r5] sums = "s1": the block above also leaves the column sums of gm; the other half of bn3's backward sums comes from this block's weight-gradient
        # GEMM (_TConv.dzfree_q_sums), so its sums pass over (gm, z3) disappears
        q = self.q_policy(eng, s["out"].shape[0])
        sums = "s1" if (q and not self.b3.frozen) else False
        if s["z3"] is None and sums != "s1":
            return None            # (z3 was not stored: only the column-sum form can do without it)
        return dict(bits=s["bits"], z3=s["z3"], bn=self.b3, sums=sums)

    def launch_sums_pair(self, a2, x, x_pitch, g, bits, m, eng):
        """Exactly one launch (bench.py brackets this call with HIP events) + the two finalizes."""
        c3, cd = self.c3, self.cd
        ns = lib.mvf_conv1x1_bwd_fused_splits(m, c3.cout, c3.cin)
        pa = eng.buf((id(c3), "bwpart_pair"), (c3.cout, 2 * ns, 2), torch.float32)
        pb = eng.buf((id(cd), "bwpart_pair"), (c3.cout, 2 * ns, 2), torch.float32)
        check(lib.mvf_conv1x1_bnbwd_sums_pair(_p(a2), c3.cin, _p(c3.wp), _p(x), x_pitch, _p(cd.wp), _p(g), c3.cout, _p(bits), m, c3.cout, c3.cin, _p(self.b3.mean),
                                              _p(self.b3.invstd), _p(self.bd.mean), _p(self.bd.invstd), _p(pa), _p(pb), 2 * ns, eng.dt, _st()), "bn backward sums (pair)")
        check(lib.mvf_bn_bwd_finalize(_p(pa), 2 * ns, c3.cout, _p(self.b3.dgamma), _p(self.b3.dbeta), _st()), "mvf_bn_bwd_finalize")
        check(lib.mvf_bn_bwd_finalize(_p(pb), 2 * ns, c3.cout, _p(self.bd.dgamma), _p(self.bd.dbeta), _st()), "mvf_bn_bwd_finalize")

    def forward(self, x, nt, h, w, c, eng):
        m = nt * h * w
        s = dict(x=x, h=h, w=w, c=c)
        # The downsample branch conv -> BN statistics only depends on the block input: it runs on the side stream (idle during the
        # forward pass) beside conv1 -> bn1 -> conv2 -> bn2 -> conv3 and is joined before bn3's apply adds the two branches.
        side = eng.side_stream() if (self.cd is not None and eng.overlap_downsample and eng.fuse_stats) else None
        if side is not None:
            eng._wait(side, eng.main_stream())
            with _on_stream(side):
                zd, _, _ = self.cd.forward(x, nt, h, w, bn=self.bd)
        if self.mvf is not None:
            s["y"], o = self.mvf.forward(x, nt, h, w, c, eng)
            if self.split_ok:
                s["o"] = o
                z1, _, _ = self.c1.forward(x, nt, h, w, c, x2=o, split_c=self.mvf.cs, bn=self.b1)
            else:   # odd slice widths: materialise [o | x_rest]
                xin = x.clone()
                xin[:, : self.mvf.cs] = o
                s["xin"] = xin
                z1, _, _ = self.c1.forward(xin, nt, h, w, c, bn=self.b1)
        else:
            z1, _, _ = self.c1.forward(x, nt, h, w, c, bn=self.b1)
        a1 = self.b1.apply(z1, m, 1)
        z2, ho, wo = self.c2.forward(a1, nt, h, w, bn=self.b2)
        m2 = nt * ho * wo
        gram = self.gram_fwd(eng, m2)
        cs = gram and eng.gram_colsums and self.b2.c % 8 == 0
        if cs:
            amean = eng.buf((id(self.c3), "amean"), (4, self.c3.cin), torch.float32)
            a2 = self.b2.apply_colmeans(z2, m2, 1, amean[0])
        else:
            a2 = self.b2.apply(z2, m2, 1)
        if gram:           # [r5] no first conv3 pass at all: bn3's statistics from the Gram matrix of a2
            self.c3.gram_stats(a2, self.b3, nt, ho, wo, eng, means_done=cs)
            z3 = None
        else:
            z3, _, _ = self.c3.forward(a2, nt, ho, wo, bn=self.b3, store=not self.z3_free(eng, m2))
        if self.cd is not None:
            if side is not None:
                eng._wait(eng.main_stream(), side)                # the downsample branch (queued before conv1, see above)
            else:
                zd, _, _ = self.cd.forward(x, nt, h, w, bn=self.bd)
            if self.fuse_apply(eng):
                out, bits = self.c3.forward_apply(a2, nt, ho, wo, self.b3, zd, self.bd)
            else:
                out, bits = self.b3.apply(z3, m2, 1, residual=zd, rbn=self.bd, bits=True)
            s["zd"] = zd
        elif self.fuse_apply(eng):
            out, bits = self.c3.forward_apply(a2, nt, ho, wo, self.b3, x)
        else:
            out, bits = self.b3.apply(z3, m2, 1, residual=x, bits=True)
        s.update(z1=z1, a1=a1, z2=z2, a2=a2, z3=z3, out=out, bits=bits, ho=ho, wo=wo, gram=gram)
        self.saved = s
        return out, ho, wo, self.c3.cout

    def backward(self, g, nt, eng, g_gated=False, gate=None, sums_done=False):
        """g: gradient of the block output; g_gated: it already IS gm = g * [out > 0] (the block above gated it for us; sums_done: and took bn3's backward
        sums).  gate: the request of the block BELOW (wants_gated_gradient) when that block wants its gradient gated the same way; self.gated_out /
        self.sums_out say what the returned dx went through."""
        out_gate = gate["bits"] if gate is not None else None
        gsum = gate if (gate is not None and gate["sums"]) else None
        s = self.saved
        h, w, c, ho, wo = s["h"], s["w"], s["c"], s["ho"], s["wo"]
        m, m2 = nt * h * w, nt * ho * wo
        bits = s["bits"]       # sign bits of the block output: the ReLU mask of g, applied wherever g is consumed (never materialised)
        self.gated_out = self.sums_out = False
        if g_gated:
            assert self.dzfree(eng, m2), "a gated gradient was produced for a block that does not take it"
        # Order matters for the two-stream overlap: each weight-gradient GEMM is queued on the side stream AFTER the
        # data-gradient GEMM that shares its dz has been queued on the main stream, so it starts when the main stream
        # moves on to the (HBM-bound) BatchNorm-backward kernels of the next layer -- MFMA work under memory work --
        # instead of fighting the data-gradient GEMM for the matrix cores.
        dzd = resid_ds = None
        w3_done = wd_done = False        # [r4] weight gradient already taken inside the BatchNorm-backward pass
        q_first = s["z3"] is None and g_gated and sums_done == "s1"       # [r5] z3 neither stored nor needed (q_z3_free)
        if self.cd is not None and s["z3"] is None:
            # [r4] z3-free downsample block: each branch = sums pass + one-pass backward on the recomputed conv; neither dz3 nor dz_d exists
            pair = bool(eng.pair_ds_sums & 1)
            if pair:           # bn3's and bn_d's sums in ONE pass over g (csrc/pw_sums_pair.hip)
                self.launch_sums_pair(s["a2"], s["x"], c, g, bits, m2, eng)
            da2 = self.c3.bwd_fused(s["a2"], g, bits, nt, ho, wo, self.b3, self.b2, s["z2"], sums_done=pair)
            resid_ds = self.cd.bwd_fused(s["x"], g, bits, nt, h, w, self.bd, a_pitch=c, sums_done=pair)
            dz3, w3_done, wd_done = None, True, True
        elif self.cd is not None and eng.pair_bn_bwd and not (self.b3.frozen or self.bd.frozen):
            if (eng.fuse_bnwg & 2) and self.c3.fuses_wgrad(eng, m2, self.c3.cout, 4, 2):
                # (mvf_bn_bwd_pair_wgrad contracts BOTH convs only with its 64-wide k tile: wider ones pass x_b = NULL and keep the GEMM)
                both = self.cd.stride == 1 and self.cd.cin == self.c3.cin and self.cd.kh == 1 and self.c3.cin == 64
                dz3, dzd = _BN.backward_pair_wgrad(self.b3, self.bd, g, self.c3.cout, s["z3"], s["zd"], m2, eng, bits, self.c3, s["a2"], self.c3.cin,
                                                   self.cd, s["x"] if both else None, c)
                w3_done, wd_done = True, both
            else:
                dz3, dzd = _BN.backward_pair(self.b3, self.bd, g, self.c3.cout, s["z3"], s["zd"], m2, eng, bits)
        elif s["z3"] is None and not q_first and eng.fuse_c3_bwd and eng.fuse_bn_bwd_sums and self.c3.bwd_fused_ok(m2):
            # [r4] z3 never stored AND dz3 never stored: one pass forms it per 64-pixel chunk and contracts it three ways
            da2 = self.c3.bwd_fused(s["a2"], g, bits, nt, ho, wo, self.b3, self.b2, s["z2"])
            dz3, w3_done = None, True
        elif s["z3"] is None and not q_first:          # z3 was never stored: bn3's backward on the recomputed conv3
            dz3 = self.c3.bwd_recompute(s["a2"], g, bits, nt, ho, wo, self.b3)
        elif self.dzfree(eng, m2):
            # [r5] no dz3: bn3's sums on gm and z3, then the data gradient on [gm | a2] (+ bn2's sums) and, on the side stream, the weight gradient on gm
            b3 = self.b3
            q_done = False
            if g_gated:
                gm = g
                if sums_done == "s1":
                    self.c3.dzfree_q_sums(gm, s["a2"], b3, nt, ho, wo, eng)
                    q_done = True
                elif not sums_done:
                    b3._reduce(gm, self.c3.cout, s["z3"], m2, eng, 0, None, None)
            else:
                assert s["z3"] is not None
                gm = eng.buf((id(b3), "gm"), s["z3"].shape, s["z3"].dtype)
                b3._reduce(g, self.c3.cout, s["z3"], m2, eng, 4, bits, gm)
            fz = b3._zero if b3.frozen else None
            da2 = self.c3.dzfree_dgrad(gm, s["a2"], b3, nt, ho, wo, self.b2, s["z2"], frozen_zero=fz)
            self.c3.dzfree_wgrad(gm, s["a2"], b3, nt, ho, wo, eng, frozen_zero=fz, q_done=q_done, gram_done=bool(s.get("gram")))
            dz3, w3_done = None, True
            del gm
        elif (eng.fuse_bnwg & 1) and self.c3.fuses_wgrad(eng, m2, self.c3.cout, 4):
            dz3 = self.b3.backward_wgrad(g, self.c3.cout, s["z3"], m2, eng, 4, bits, self.c3, s["a2"], self.c3.cin)
            w3_done = True
        else:
            dz3 = self.b3.backward(g, self.c3.cout, s["z3"], m2, eng, 4, ymask=bits)
        eng.issue_marked()         # [r5] the block above's side launches, enqueued behind this block's first launch-stream kernels (see side_late)
        fuse = eng.fuse_bn_bwd_sums
        if dz3 is None:
            pass       # the fused pass above already produced da2 and bn2's sums
        elif fuse:     # the data gradient's epilogue also produces the BatchNorm-backward sums of the BN its output feeds
            da2 = self.c3.dgrad_bnsums(dz3, nt, ho, wo, ho, wo, self.b2, s["z2"])
        else:
            da2 = self.c3.dgrad(dz3, nt, ho, wo, ho, wo)
        if not w3_done:
            self.c3.wgrad(dz3, s["a2"], nt, ho, wo, ho, wo, eng)
        del dz3
        dz2 = self.b2.backward(da2, self.c2.cout, s["z2"], m2, eng, 2, sums_done=fuse)
        del da2
        fuse1 = fuse and (self.c2.stride == 1 or eng.fuse_bn_bwd_strided)
        if fuse1:
            da1 = self.c2.dgrad_bnsums(dz2, nt, ho, wo, h, w, self.b1, s["z1"])
        else:
            da1 = self.c2.dgrad(dz2, nt, ho, wo, h, w)
        self.c2.wgrad(dz2, s["a1"], nt, h, w, ho, wo, eng)
        del dz2
        w1_done = (eng.fuse_bnwg & 4) and self.mvf is None and self.c1.fuses_wgrad(eng, m, self.c1.cout, 2)
        if w1_done:
            dz1 = self.b1.backward_wgrad(da1, self.c1.cout, s["z1"], m, eng, 2, None, self.c1, s["x"], c, sums_done=fuse1)
        else:
            dz1 = self.b1.backward(da1, self.c1.cout, s["z1"], m, eng, 2, sums_done=fuse1)
        del da1
        resid, rbits = (g, None) if g_gated else (g, bits)
        if resid_ds is not None:
            resid, rbits = resid_ds, None
        elif self.cd is not None:
            if dzd is None:
                dzd = self.bd.backward(g, self.cd.cout, s["zd"], m2, eng, 4, ymask=bits)
            resid, rbits = self.cd.dgrad(dzd, nt, ho, wo, h, w), None
            if not wd_done:
                self.cd.wgrad(dzd, s["x"], nt, h, w, ho, wo, eng, x_pitch=c)
            del dzd
        if self.mvf is None:
            dx = self.c1.dgrad(dz1, nt, h, w, h, w, residual=resid, res_bits=rbits, out_gate=out_gate, gsum=gsum)
            self.gated_out, self.sums_out = out_gate is not None, (gsum["sums"] if gsum is not None else False)
            if not w1_done:
                self.c1.wgrad(dz1, s["x"], nt, h, w, h, w, eng, x_pitch=c)
        else:
            # skip-connection gradient without a separate add pass: the data-gradient epilogue adds it to the pass-through
            # channels (>= cs); the slice [0, cs) first goes back through the MVF, whose transposed stencil adds its share
            fuse = self.mvf.cs % 4 == 0
            og = out_gate if fuse else None
            gs = gsum if fuse else None
            dxp = self.c1.dgrad(dz1, nt, h, w, h, w, residual=resid if fuse else None, res_c0=self.mvf.cs, res_bits=rbits if fuse else None, out_gate=og, gsum=gs)
            self.gated_out, self.sums_out = og is not None, (gs["sums"] if gs is not None else False)
            if self.split_ok:
                self.c1.wgrad(dz1, s["x"], nt, h, w, h, w, eng, x_pitch=c, x2=s["o"], split_c=self.mvf.cs)
            else:
                self.c1.wgrad(dz1, s["xin"], nt, h, w, h, w, eng, x_pitch=c)
            self.mvf.backward(dxp, s["x"], s["y"], nt, h, w, c, eng, addend=resid if fuse else None, addend_bits=rbits if fuse else None, out_gate=og, gsum=gs)
            if not fuse:
                if rbits is not None:       # odd slice widths: materialise g * mask once
                    resid = resid * (s["out"] > 0).to(resid.dtype)
                dx = eng.add(dxp, resid, key=id(self))
            else:
                dx = dxp
        if eng.keep_io:      # parity tests: this block's boundary tensors of the step (persistent buffers, valid until the next step)
            self.io = dict(x=s["x"], out=s["out"], g=g, dx=dx, h=h, w=w, c=c, ho=ho, wo=wo, g_gated=g_gated, dx_gated=self.gated_out)
        self.saved = None
        eng.flush_side(late=True)
        return dx


class _ParamStore(object):
    """Flat fp32 parameter / gradient / momentum buffers for a module; its nn.Parameters become views."""

    def _init_store(self, model, dtype=torch.float32, rehome=True):
        """rehome=True: the parameters become views of this store's flat buffer (the optimizer kernel and the gradient exchange work on
        it).  rehome=False (Bottleneck.forward as a stand-alone autograd node): the parameters stay where they are -- possibly views of
        a MODEL engine's flat buffer, which must keep aliasing them -- and only the gradient buffer is private; every kernel reads the
        parameters through the module's own tensors, so nothing is copied; apply_sgd is not available on such a store."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise TypeError("training dtype must be float32 or bfloat16 (activation storage; weights/statistics/gradients stay fp32)")
        self.tdtype = dtype
        self.dt = _lib.MVF_F32 if dtype == torch.float32 else _lib.MVF_BF16
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("the HIP training engine needs the model on an MI355X device; no CPU fallback")
        self.model, self.device = model, dev
        self._side_pending = []
        self._side_keep = []
        params = [p for p in model.parameters()]
        n = sum(p.numel() for p in params)
        pad = lambda k: (k + 3) // 4 * 4
        total = sum(pad(p.numel()) for p in params)
        self.rehomed = bool(rehome)
        self.flat_params = torch.zeros(total, device=dev) if rehome else None
        self.flat_grads = torch.zeros(total, device=dev)
        self.flat_mom = torch.zeros(total, device=dev) if rehome else None
        self._grad_view = {}
        off = 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                if rehome:
                    v = self.flat_params[off:off + k].view(p.shape)
                    v.copy_(p.data.float())
                    p.data = v
                elif p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                    raise TypeError("stand-alone block: parameters must be contiguous float32 tensors on %s" % dev)
                gv = self.flat_grads[off:off + k].view(p.shape)
                self._grad_view[id(p)] = gv
                off += pad(k)
        self.n_params = n
        if rehome:
            for m_ in model.modules():                  # cached inference engines / block trainers hold (packed copies of) the old storage
                if hasattr(m_, "invalidate_engine"):
                    m_.invalidate_engine()
        self._ws = None
        self.norm_out = torch.zeros(2, device=dev)
        self.steps = 0
        self._ones = {}
        self._bufs = {}
        self._caps = {}
        self._last_shape = {}

    def grad_of(self, p):
        return self._grad_view[id(p)]

    def buf(self, key, shape, dtype=None):
        """Persistent per-call-site device buffer (activations, gradients, scratch): shapes are static from step to step, so
        nothing goes through the caching allocator inside a step (torch.empty + record_stream bookkeeping cost ~60 us per
        call here and made the host the bottleneck at 36 ms/step)."""
        dt = dtype or self.tdtype
        shape = tuple(shape)
        k = (key, shape, dt)
        # [r5] every shape asked for under one key is a view of ONE allocation: two shapes inside one step would alias silently
        step = getattr(self, "forward_count", None)
        if step is not None:
            last = self._last_shape.get((key, dt))
            if last is not None and last[0] == step and last[1] != shape:
                raise RuntimeError("engine buffer %r requested with shapes %s and %s inside one step: they would share storage" % (key, last[1], shape))
            self._last_shape[(key, dt)] = (step, shape)
        t = self._bufs.get(k)
        if t is None:
            # One allocation per call site, sized for the LARGEST shape seen there; other shapes (the partial last batch of an epoch:
            # drop_last=False as in the reference) are views of it instead of a second resident set of activation buffers.
            n = 1
            for v in shape:
                n *= int(v)
            ck = (key, dt)
            flat = self._caps.get(ck)
            if flat is None or flat.numel() < n:
                if flat is not None:                   # a larger batch arrived: side-stream kernels may still read the old storage
                    self._side_keep.append(flat)
                    self._storage_epoch += 1           # launch plans recorded so far hold pointers into the old storage
                    for kk in [kk for kk in self._bufs if kk[0] == key and kk[2] == dt]:
                        del self._bufs[kk]
                flat = torch.empty(max(n, 1), device=self.device, dtype=dt)
                self._caps[ck] = flat
            t = flat[:n].view(shape)
            self._bufs[k] = t
        return t

    def workspace(self, nbytes, side=False):
        key = "_ws_side" if side else "_ws"
        ws = getattr(self, key, None)
        if ws is None or ws.numel() < nbytes:
            if ws is not None:
                self._storage_epoch += 1               # (as in buf(): recorded launch plans point at the old workspace)
            ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=self.device)
            setattr(self, key, ws)
        return ws

    _storage_epoch = 0         # bumped whenever a persistent buffer or workspace is REPLACED by a larger one: a launch plan is valid for the epoch it was recorded in

    # ---- policy.  The defaults below ARE the measured policy of the bf16 / fp32 train step (DESIGN.md section 4.5 has the table with the measurement behind
    # every line); MVF_POLICY="name=value,name=value" (read once, _policy below) overrides them for A/B runs, tests set the attributes on an engine.
    overlap_wgrad = True       # weight gradients / tap gradients / slab reduces on a second stream
    keep_io = False            # keep references to every block's input / output / gradients after backward (teacher-forced parity tests)
    fuse_stats = True          # BatchNorm batch statistics accumulated in the producing conv's epilogue
    overlap_downsample = _policy("side_downsample", 1) != 0        # forward: the downsample branch's conv on the side stream
    z3_free = _policy("z3_free", 1)                    # [r3] plain blocks with the fused apply: z3 never stored (0 off, 1 planes <= 64, 2 all of them)
    fuse_bn3_apply = _policy("fuse_bn3_apply", 1)      # [r3] bn3 apply + residual + ReLU as a second conv3 pass (0 never / 1 planes <= 128 / 2 every block)
    fuse_bn_bwd_sums = _policy("fuse_bn_bwd", 1) != 0  # BatchNorm-backward sums in the data gradient's epilogue
    fuse_bn_bwd_strided = _policy("fuse_bn_bwd_strided", 1) != 0      # ... also for the stride-2 convs' parity classes
    pair_bn_bwd = _policy("pair_bn_bwd", 1) != 0       # downsample blocks: bn3 + downsample-BN backward in one pass over g
    # [r4] weight gradients of layer1 / layer2's pointwise convs inside the BatchNorm-backward apply pass that forms their dz (bit mask:
    # 1 conv3 of plain stored-z3 blocks, 2 conv3 (+ stride-1 downsample conv) of downsample blocks, 4 conv1 of blocks without MVF,
    # 8 also give up the z3-free path of layer1's plain blocks for it)
    fuse_bnwg = _policy("fuse_bnwg", 7)
    pair_ds_sums = _policy("pair_ds_sums", 3)          # [r4] bit 0: layer1.0's bn3 and bn_d backward sums in one pass over g (csrc/pw_sums_pair.hip); bit 1: the plain z3-free blocks' sums on its one-branch form
    z3_free_ds = _policy("z3_free_ds", 1)              # [r4] layer1.0 (downsample block, both convs 64 -> 256 pointwise): no stored z3, per-branch sums + one-pass backward
    fuse_c3_bwd = _policy("fuse_c3_bwd", 1)            # [r4] z3-free blocks: conv3 recompute + bn3 backward apply + data gradient (+ bn2 sums) + weight gradient in ONE pass (csrc/pw_bwd_fused.hip)
    fuse_mvf_stats = _policy("fuse_mvf_stats", 1) != 0  # [r5] MVF's BatchNorm statistics accumulated by the stencil launch (0 = a pass over y)
    # [r5] bn3's backward of plain blocks without the dz3 tensor (csrc/bn_dzfree.hip): 0 off, 1 planes >= 256, 2 every eligible block (layer2 ... layer4's plain
    # blocks); gate_producer: the block above gates the gradient it hands down (0 = the dz3-free block's own sums pass writes gm)
    dzfree = _policy("dzfree", 2)                      # (measured in the step, one box: C3 19.13 -> 19.03 (1) / 18.78 ms (2); C4 33.66 -> 32.96 / 32.68 ms)
    gate_producer = _policy("gate_producer", 1) != 0
    # [r5] dzfree_q: bn3's backward sums of a dz3-free block without ANY pass over (gm, z3): the block above leaves the column sums of gm (its epilogues) and
    # sum gm z3 = sum_k W Q comes from the weight-gradient GEMM Q = gm^T a2, which moves from the side stream to the launch stream, ahead of the data gradient
    # (dzfree_q_wgs workgroups: the launch stream waits for it, so it takes the whole chip).  0 off, 1 planes <= dzfree_q_maxk and >= 160 MB of (gm, z3), 2 every
    # dz3-free block.  Measured in the step (ms, 0 / 1-without-the-size-rule / 2): C4 32.31-32.46 / 31.62-31.91 / 31.68-31.81; C3 18.52-18.66 / 18.54-18.67 /
    # 18.52-18.68; 12 clips 9.37-9.41 / 9.55-9.57 (launch-bound: two more launches per block cost more than the short pass) -- hence the size rule.
    dzfree_q = _policy("dzfree_q", 1)
    dzfree_q_wgs = _policy("dzfree_q_wgs", 256)
    dzfree_q_maxk = _policy("dzfree_q_maxk", 256)
    dzfree_q_mink = _policy("dzfree_q_mink", 0)
    # [r5] gram_stats: bn3's batch statistics of a block that applies bn3 in a second conv3 pass from the Gram matrix of a2 (_TBlock.gram_fwd) -- the first
    # conv3 pass (statistics only in layer1, z3-storing in layer2) disappears.  gram_stats_wgs: the workgroup count of that GEMM on the launch stream;
    # gram_colsums: the column means of a2 from the bn2 apply that writes it (mvf_bn_apply_colmeans) instead of a pass over a2
    gram_stats = _policy("gram_stats", 1) != 0
    gram_stats_wgs = _policy("gram_stats_wgs", 256)
    gram_colsums = _policy("gram_colsums", 1) != 0
    # ([r6] removed after two rounds at "measured neutral or slower", records in profiles/r05_*: the complete bn3 sums in the gating epilogues (gate_sums), Q's slabs
    # summed by the sums kernel, the statistics-only conv3 pass of dz3-free blocks without the Gram form, the Gram form for layer1.0 and for planes = 256, a third
    # stream for the downsample branch's data gradient, holding the last stages' weight gradients back, per-launch / twice-per-block side hand-overs, the
    # scatter forms of the stem's pooling backward)

    def side_stream(self):
        if not self.overlap_wgrad:
            return None
        if getattr(self, "_side", None) is None:
            from .streams import concurrent_stream
            self._side = concurrent_stream(self.main_stream())      # not every new stream gets its own hardware queue (streams.py)
        return self._side

    def main_stream(self):
        ms = getattr(self, "_main", None)
        return ms if ms is not None else torch.cuda.current_stream()

    # ---- cross-stream orderings: through these three, so that a step being recorded into a launch plan (launch_plan.Recorder) sees them
    _rec = None

    def _wait(self, dst, src):
        """dst waits for everything queued on src so far."""
        if self._rec is not None:
            self._rec.wait(dst, src)
        else:
            dst.wait_stream(src)

    def _mark(self, stream):
        if self._rec is not None:
            return self._rec.mark(stream)
        ev = torch.cuda.Event()
        ev.record(stream)
        return ev

    def _wait_mark(self, stream, ev):
        if self._rec is not None:
            self._rec.wait_mark(stream, ev)
        else:
            stream.wait_event(ev)

    def on_side(self, launch):
        """Run `launch` (which enqueues kernels through _st()) on the side stream, ordered after everything queued on the main
        stream so far.  Each cross-stream hand-over costs a ~7 us bubble on the launch queue, so the launches of a block are collected and
        flush_side() issues them behind ONE wait ([r2]: one per side launch 23.55 ms, one per block 23.14, two per block 23.26)."""
        # the closure (and with it every tensor the side kernels read or write: dz, activations, the side workspace of the moment)
        # stays referenced until join_side() has ordered the launch stream behind the side stream: tensors are allocated on the
        # launch stream, so dropping the last reference earlier lets the caching allocator hand their memory to the next
        # launch-stream allocation while the side kernels are still using it (seen as wrong gradients in the FIRST step of an
        # engine, when buffers are still being created, once hand-overs were batched per block)
        self._side_keep.append(launch)
        self._side_pending.append(launch)

    # [r5] side_late: the hand-over is MARKED at the block's end (an event on the launch stream) but the side launches are ENQUEUED only after the next
    # block's first launch-stream kernels -- host order only, the device-side dependencies are the same.  Why: the side stream's first kernel after a hand-over
    # is a 256 x 256 weight-gradient tile (512 threads x 256 registers: a workgroup owns its CU's whole register file), and with ~14 side launches between the
    # event and the next launch-stream kernel in HOST order that kernel reached its queue after the tile had taken every CU: the launch queue then idled 21-51 us
    # at every MVF block boundary once the dz3-free path had lengthened the side list (tools/gap_dump.py, profiles/r05_step_timeline_gaps.txt): 0.25 ms of queue
    # idle per step.  With the late enqueue the gaps are back at the 7 us of a hand-over (main-queue idle 0.45 -> 0.20 ms per step) -- and the STEP does not move
    # (19.27 vs 19.27 ms, C4 33.63 vs 33.63, 12 clips 10.00 vs 10.00: three alternations): the two queues share the chip work-conservingly, the kernel that
    # started late had only been waiting for CUs the other queue was using.  Kept for the cleaner timeline.

    def flush_side(self, late=False):
        pend = getattr(self, "_side_pending", None)
        marked = getattr(self, "_side_marked", None)
        if late and pend:
            # mark now, enqueue later (issue_marked)
            ev = self._mark(self.main_stream())
            if marked is None:
                marked = self._side_marked = []
            marked.append((ev, list(pend)))
            del pend[:]
            return
        self.issue_marked()
        if pend:
            self._wait(self._side, self.main_stream())
            with _on_stream(self._side):
                for launch in pend:
                    launch()
            del pend[:]

    def issue_marked(self):
        marked = getattr(self, "_side_marked", None)
        if marked:
            for ev, launches in marked:
                self._wait_mark(self._side, ev)
                with _on_stream(self._side):
                    for launch in launches:
                        launch()
            del marked[:]

    def join_side(self):
        self.flush_side()
        if getattr(self, "_side", None) is not None:
            self._wait(self.main_stream(), self._side)
        del self._side_keep[:]                      # from here on the launch stream is ordered behind every side kernel

    def add(self, a, b, key=None):
        """a + b (elementwise) through mvf_bn_apply with unit scale / zero shift."""
        m, c = a.shape
        if c not in self._ones:
            self._ones[c] = (torch.ones(c, device=a.device), torch.zeros(c, device=a.device))
        one, zero = self._ones[c]
        out = self.buf((key or id(a), "add"), a.shape, a.dtype) if key is not None else torch.empty_like(a)
        check(lib.mvf_bn_apply(_p(a), m, c, _p(one), _p(zero), _p(b), None, None, 0, _p(out), self.dt, _st()), "add")
        return out

    lr, momentum, weight_decay, max_norm = 0.015, 0.9, 1e-4, 40.0

    def trainable_offset(self):
        """Flat-buffer offset of the first trainable parameter.  A PREFIX of model.parameters() excluded from training
        (requires_grad False: the reference's frozen_stages, resnet.py:515-527 -- stem, then layer1..k) is simply cut off: the
        optimizer (norm, clip, weight decay, momentum, update) runs on the rest of the flat buffers, as torch's clip_grad_norm_ /
        SGD skip parameters without a gradient.  Exclusions in scattered places behind it (norm_frozen: every BatchNorm weight /
        bias, partial_norm; resnet.py:496-513) are handled by the segment form of the optimizer kernel (_segments: lr_mult -1)."""
        params = list(self.model.parameters())
        flags = [p.requires_grad for p in params]
        k = flags.index(True) if True in flags else len(flags)
        self._scattered_frozen = not all(flags[k:])
        return self._grad_view[id(params[k])].storage_offset() if k < len(params) else self.flat_grads.numel()

    nesterov = True
    param_options = None       # {id(param): (lr_mult, decay_mult)}: build_optimizer's paramwise_options (see set_param_options)

    def set_param_options(self, options):
        """Per-parameter learning-rate / weight-decay multipliers: {parameter: (lr_mult, decay_mult)} (missing = (1, 1)).
        The optimizer then runs the segment form of the fused kernel (mvf_sgd_step_segments) on the same flat buffers."""
        self.param_options = {id(p): (float(a), float(b)) for p, (a, b) in options.items()} if options else None
        self._seg_tables = {}

    def _segments(self, off):
        """Device table of optimizer segments for the flat range [off, end): neighbours with equal (lr_mult, decay_mult) share a
        segment; a parameter with requires_grad False gets lr_mult -1 = excluded (norm, update and momentum skip it)."""
        tabs = self.__dict__.setdefault("_seg_tables", {})
        key = (off, tuple(p.requires_grad for p in self.model.parameters()))
        if key not in tabs:
            segs, last = [], None
            for p in self.model.parameters():
                first = self._grad_view[id(p)].storage_offset()
                if first + p.numel() <= off:
                    continue
                mult = (self.param_options or {}).get(id(p), (1.0, 1.0)) if p.requires_grad else (-1.0, 0.0)
                if mult != last:                    # (padding elements between parameters are zeros: they may join either side)
                    segs.append(_lib.SgdSegment(max(first - off, 0) if segs else 0, mult[0], mult[1]))
                    last = mult
            arr = (_lib.SgdSegment * len(segs))(*segs)
            tabs[key] = (torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device), len(segs))
        return tabs[key]

    def apply_sgd(self, lr=None, world=1):
        """clip_grad_norm_(max_norm) + SGD on the flat buffers (gradient scaled by 1 / world): one fused launch sequence."""
        if not self.rehomed:
            raise RuntimeError("apply_sgd: this store does not own its parameters (stand-alone Bottleneck.forward); use the model's train engine or a torch optimizer")
        off = self.trainable_offset()
        n = self.flat_params.numel() - off
        if n <= 0:
            return None
        ws = self.workspace(lib.mvf_sgd_workspace_bytes(n))
        args = (_p(self.flat_params[off:]), _p(self.flat_grads[off:]), _p(self.flat_mom[off:]), n, C.c_float(1.0 / world),
                C.c_float(self.max_norm or 0.0), C.c_float(self.lr if lr is None else lr), C.c_float(self.momentum),
                C.c_float(self.weight_decay), int(self.steps == 0))
        if self.nesterov and not self.param_options and not self._scattered_frozen:
            check(lib.mvf_sgd_nesterov_step(*(args + (_p(self.norm_out), _p(ws), ws.numel(), _st()))), "sgd step")
        else:
            tab, nseg = self._segments(off)
            check(lib.mvf_sgd_step_segments(*(args + (int(self.nesterov), _p(tab), nseg, _p(self.norm_out), _p(ws), ws.numel(), _st()))), "sgd step")
        self.steps += 1
        return self.norm_out

    # ---- optimizer state in torch.optim.SGD's wire format (reference checkpoints: codes/utils/checkpoint.py:235-265) ---------
    def optimizer_state_dict(self):
        """What torch.optim.SGD(model.parameters(), ...).state_dict() would hold after the same steps: per-parameter
        `momentum_buffer` in model.parameters() order + one param group -- so `{meta, state_dict, optimizer}` checkpoints written
        here resume under the reference's runner (optimizer.load_state_dict) and the other way round."""
        from .checkpoint import sgd_state_dict
        params = list(self.model.parameters())
        off = self.trainable_offset()
        bufs = []
        for p in params:
            v = self._grad_view[id(p)]
            first = v.storage_offset()
            stepped = self.steps > 0 and first >= off and p.requires_grad
            bufs.append(self.flat_mom[first:first + p.numel()].view(p.shape).detach().cpu().clone() if stepped else None)
        mult = [self.param_options.get(id(p), (1.0, 1.0)) for p in params] if self.param_options else None
        return sgd_state_dict(bufs, self.lr, self.momentum, self.weight_decay, self.nesterov, multipliers=mult,
                              initial_lr=getattr(self, "initial_lr", None))

    def load_optimizer_state_dict(self, opt):
        """Accepts torch.optim.SGD's state_dict (the reference's checkpoints) or round 1's {'momentum_buffer': flat, 'steps': n}."""
        from .checkpoint import sgd_momentum_buffers
        if "momentum_buffer" in opt and "state" not in opt:          # round-1 layout of this repo
            self.flat_mom.copy_(opt["momentum_buffer"].to(self.flat_mom.device))
            self.steps = int(opt.get("steps", 1))
            return
        params = list(self.model.parameters())
        bufs, group = sgd_momentum_buffers(opt, len(params))
        self.flat_mom.zero_()
        any_buf = False
        for p, b in zip(params, bufs):
            if b is None:
                continue
            if tuple(b.shape) != tuple(p.shape):
                raise ValueError("optimizer state: momentum_buffer %s does not match parameter %s" % (tuple(b.shape), tuple(p.shape)))
            first = self._grad_view[id(p)].storage_offset()
            self.flat_mom[first:first + p.numel()].view(p.shape).copy_(b.to(device=self.device, dtype=torch.float32))
            any_buf = True
        for k in ("momentum", "nesterov") + (() if len(opt["param_groups"]) > 1 else ("lr", "weight_decay")):
            if k in group:          # (per-parameter groups carry multiplied lr / weight_decay values: the base values stay the engine's)
                setattr(self, k, group[k])
        self.steps = max(self.steps, 1) if any_buf else 0          # torch creates a buffer at a parameter's first step

    def attach_grads(self):
        """Expose the flat gradient views as .grad of the parameters (for external optimizers / inspection)."""
        for p in self.model.parameters():
            p.grad = self._grad_view[id(p)]


class BlockTrainer(_ParamStore):
    """Train-mode forward/backward of ONE mvfnet_amd Bottleneck (with or without MVF), or of a torch.nn.Sequential of them chained the way
    TrainEngine._backward chains a stage (gated hand-down, column sums) -- used by the parity tests."""

    def __init__(self, block, dtype=torch.float32, rehome=True):
        self._init_store(block, dtype, rehome)
        self.blks = [_TBlock(b, self) for b in block] if isinstance(block, torch.nn.Sequential) else [_TBlock(block, self)]
        self.blk = self.blks[0]
        for lo, hi in zip(self.blks[:-1], self.blks[1:]):
            lo.above = hi

    def forward(self, x_nchw):
        nt, c, h, w = x_nchw.shape
        self.nt = nt
        self._main = torch.cuda.current_stream()
        out = x_nchw.permute(0, 2, 3, 1).contiguous().view(nt * h * w, c).to(self.tdtype)
        ho, wo, co = h, w, c
        with _on_stream(self._main, main=True):
            for b in self.blks:
                for cv in b.convs():
                    cv.pack()
            for b in self.blks:
                out, ho, wo, co = b.forward(out, nt, ho, wo, co, self)
        return out.view(nt, ho, wo, co).permute(0, 3, 1, 2)

    def backward(self, g_nchw):
        nt, co, ho, wo = g_nchw.shape
        s = self.blk.saved
        h, w, c = s["h"], s["w"], s["c"]
        g = g_nchw.permute(0, 2, 3, 1).contiguous().view(nt * ho * wo, co).to(self.tdtype)
        gated = sums = False
        self._main = torch.cuda.current_stream()
        with _on_stream(self._main, main=True):
            for i in range(len(self.blks) - 1, -1, -1):
                req = self.blks[i - 1].wants_gated_gradient(self) if i > 0 else None
                g = self.blks[i].backward(g, nt, self, g_gated=gated, gate=req, sums_done=sums)
                gated, sums = self.blks[i].gated_out, self.blks[i].sums_out
            dx = g
            self.join_side()
        return dx.view(nt, h, w, c).permute(0, 3, 1, 2)


class TrainEngine(_ParamStore):
    """One-GPU training step for a mvfnet_amd Recognizer2D (fp32)."""

    def __init__(self, model, lr=0.015, momentum=0.9, weight_decay=1e-4, max_norm=40.0, dtype=torch.float32):
        """dtype = storage type of activations and packed weights (float32 | bfloat16).  With bfloat16 the accumulation,
        BatchNorm statistics, all parameter gradients, the master weights and the optimizer stay fp32 (what the
        reference's own fp16 mode keeps in fp32, codes/core/fp16/hooks.py:12-136; no loss scaling is needed for bf16)."""
        self._init_store(model, dtype)
        self.lr, self.momentum, self.weight_decay, self.max_norm = lr, momentum, weight_decay, max_norm
        bb = model.backbone
        self.stem, self.stem_bn = _TConv(bb.conv1, self, stem=True), _BN(bb.bn1, self, "bn1")
        self.blocks = [_TBlock(blk, self) for name in bb.res_layers for blk in getattr(bb, name)]
        for lo, hi in zip(self.blocks[:-1], self.blocks[1:]):
            lo.above = hi
        head = model.cls_head
        self.fc_w, self.fc_b = head.new_fc.weight, head.new_fc.bias
        self.dfc_w, self.dfc_b = self.grad_of(head.new_fc.weight), self.grad_of(head.new_fc.bias)
        self.dropout = head.dropout_ratio if head.dropout is not None else 0.0
        self.num_classes = head.num_classes
        # Gradient exchange in two buckets (see _launch_tail_allreduce): the flat buffer follows model.parameters() order (stem,
        # layer1..4, head), backward produces it back to front, so the slice from the first parameter of the third residual
        # stage to the end (layer3 + layer4 + fc = 94 % of ResNet-50) is complete when backward enters layer2.
        self._tail_off, self._tail_block = None, None
        stages_ = [getattr(bb, name) for name in bb.res_layers]
        if len(stages_) >= 3:
            first = next(stages_[2].parameters())
            self._tail_off = self._grad_view[id(first)].storage_offset()
            self._tail_block = sum(len(st_) for st_ in stages_[:2])        # index of the first block of that stage
        self._works = []
        self._tail_launched = False
        # every BatchNorm's num_batches_tracked becomes a 0-dim view of one int64 buffer (see _BN._count)
        bns = [m_ for m_ in model.modules() if isinstance(m_, torch.nn.modules.batchnorm._BatchNorm) and m_.num_batches_tracked is not None]
        self._nbt_flat = torch.zeros(max(len(bns), 1), dtype=torch.int64, device=self.device)
        for i, m_ in enumerate(bns):
            self._nbt_flat[i] = m_.num_batches_tracked.to(self.device)
            m_.num_batches_tracked = self._nbt_flat[i]
        self.forward_count = 0
        self.saved = None
        self._nbt_touched = False
        self._nbt_mods, self._nbt_inc = bns, {}      # increment vectors per pattern of training / eval BatchNorms (frozen ones keep theirs)
        # uint8 input path (preprocess.FramePipeline): decoded frames in, crop / flip / normalise fused into the stem prep;
        # input_window = per-frame (y0, x0, flip) rows for the NEXT forward (None = top-left window, no flip)
        self.input_pipeline, self.input_window = None, None
        self._pack_tables = None

    def _pack_all(self, kind):
        """All forward (kind 0) or data-gradient (kind 1) weight packs in ONE launch (mvf_pack_conv_weights_batched); the job
        table is built once -- parameters and packed operands are persistent buffers."""
        if self._pack_tables is None:
            self._pack_tables = {}
            convs = [self.stem] + [cv for blk in self.blocks for cv in blk.convs()]
            for k in (0, 1):
                jobs, first = [], 0
                for cv in convs:
                    j = cv.pack_jobs()[k]
                    if j is None:
                        continue
                    w, out, cout, cin, kh, kw, kwp, cinp, kd = j
                    total = cout * kh * kwp * cinp if kd == 0 else cin * kh * kw * cout
                    tiled = cout % 32 == 0 and cin % 32 == 0 and kh * kw <= 9 and kwp == kw and cinp == cin     # LDS-tiled transpose (kinds 2 / 3)
                    jobs.append(_lib.PackJob(w.data_ptr(), out.data_ptr(), cout, cin, kh, kw, kwp, cinp, kd + 2 if tiled else kd, first))
                    first += (cout // 32) * (cin // 32) if tiled else (total + 2047) // 2048
                if not jobs:
                    self._pack_tables[k] = None
                    continue
                arr = (_lib.PackJob * len(jobs))(*jobs)
                host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
                self._pack_tables[k] = (host.to(self.device), len(jobs), first)
        t = self._pack_tables[kind]
        if t is not None:
            check(lib.mvf_pack_conv_weights_batched(_p(t[0]), t[1], t[2], self.dt, _st()), "pack_conv_weights_batched")

    # ---- one step -----------------------------------------------------------------------------------------------
    def forward(self, imgs, labels, stages=None, _prepared=None):
        """imgs [B, T, 3, H, W] fp32 -- or decoded frames [B, T, Hs, Ws, 3] uint8 with `input_pipeline` set --, labels [B, 1] /
        [B] int64 (GPU) -> loss tensor (1,), keeps activations."""
        if not imgs.is_cuda or imgs.dtype not in (torch.float32, torch.uint8):
            raise RuntimeError("TrainEngine.forward: float32 (or uint8 frames) GPU input required")
        self._main = torch.cuda.current_stream()
        self.forward_count += 1
        with _on_stream(self._main, main=True):
            return self._forward(imgs, labels, stages, _prepared)

    def _step_tensors(self, imgs, labels):
        """The three tensors of a step that are made in Python: the flat int64 labels, the dropout mask over the pooled [frames, channels] features
        (nn.Dropout of the reference head, tsn_clshead.py: F.dropout of a ones tensor = the same draw, one launch) and the fresh loss tensor."""
        lab = labels.reshape(-1).to(device=imgs.device, dtype=torch.int64).contiguous()
        mask = None
        if self.dropout > 0.0:
            nt, cc = imgs.shape[0] * imgs.shape[1], self.fc_w.shape[1]
            cache = self.__dict__.setdefault("_drop_ones", {})
            ones = cache.get((nt, cc))
            if ones is None:
                ones = cache[(nt, cc)] = torch.ones(nt, cc, device=imgs.device, dtype=torch.float32)
            mask = torch.nn.functional.dropout(ones, self.dropout, True)
        # a fresh tensor per step: the caller keeps it (a persistent buffer + clone() went through hipMemcpyAsync: a blit on another queue = ~70 us of idle
        # launch stream per step)
        loss = torch.empty(1, device=imgs.device, dtype=torch.float32)
        return lab, mask, loss

    def set_options(self, lr=None, momentum=None, weight_decay=None, max_norm=None, dtype=None):
        """Update the optimizer hyper-parameters of an existing engine (Recognizer2D.train_engine(**opt) on a model that already
        has one); the storage dtype is fixed at construction."""
        if dtype is not None and dtype != self.tdtype:
            raise RuntimeError("the model's train engine was built with dtype %s; it cannot be switched to %s" % (self.tdtype, dtype))
        for k, v in (("lr", lr), ("momentum", momentum), ("weight_decay", weight_decay), ("max_norm", max_norm)):
            if v is not None:
                setattr(self, k, v)

    def _forward(self, imgs, labels, stages=None, prepared=None):
        b, t = imgs.shape[0], imgs.shape[1]
        lab, mask, loss = prepared if prepared is not None else self._step_tensors(imgs, labels)
        u8 = imgs.dtype == torch.uint8               # decoded (B, T, Hs, Ws, 3) frames: crop/flip/normalise fused into the stem prep
        if u8:
            if self.input_pipeline is None:
                raise RuntimeError("uint8 frames need engine.input_pipeline = preprocess.FramePipeline(...)")
            x = None
            nt = b * t
            h, w = self.input_pipeline.crop_hw
        else:
            x = imgs.reshape((-1, 3) + tuple(imgs.shape[3:])).contiguous()
            nt, _, h, w = x.shape
        self._pack_all(0)                                  # every forward weight pack, one launch
        # the data-gradient packs are not needed before backward: off the critical path, on the side stream
        side = self.side_stream()
        if side is not None:
            self._wait(side, self.main_stream())           # after the previous step's parameter update
        with _on_stream(side if side is not None else self.main_stream()):
            self._pack_all(1)
        self._packs_on_side = side is not None
        hp, wp = h + 6, (w + 6 + 2 + 1) // 2 * 2
        xp = self.buf("xp", (nt, hp, wp, 4))
        if u8:
            self.input_pipeline.to_stem(imgs, self.input_window, 3, wp, self.tdtype, out=xp)
        else:
            check(lib.mvf_stem_prep(_p(x), nt, 3, h, w, 3, wp, _p(xp), self.dt, _st()), "stem_prep")
        ho, wo = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
        z0, _, _ = self.stem.forward(xp, nt, hp, wp, ho=ho, wo=wo, bn=self.stem_bn)
        h2, w2 = (ho - 1) // 2 + 1, (wo - 1) // 2 + 1
        p0 = self.buf("p0", (nt * h2 * w2, 64))
        amax = self.buf("amax", (nt * h2 * w2, 64), torch.uint8)
        check(lib.mvf_maxpool_bn_relu_fwd(_p(z0), nt, ho, wo, 64, _p(self.stem_bn.scale), _p(self.stem_bn.shift), _p(p0), _p(amax), self.dt, _st()), "maxpool fwd")
        self.saved = dict(xp=xp, z0=z0, amax=amax, nt=nt, hp=hp, wp=wp, ho=ho, wo=wo, t=t, b=b)
        if stages is not None:
            stages["maxpool"] = p0.view(nt, h2, w2, 64)
        xcur, hc, wc, cc = p0, h2, w2, 64
        ends, k = [], 0
        for name in self.model.backbone.res_layers:
            k += len(getattr(self.model.backbone, name))
            ends.append(k)
        for i, blk in enumerate(self.blocks):
            xcur, hc, wc, cc = blk.forward(xcur, nt, hc, wc, cc, self)
            if stages is not None and (i + 1) in ends:
                stages["layer%d" % (ends.index(i + 1) + 1)] = xcur.view(nt, hc, wc, cc)
        # head + loss
        dev = imgs.device
        f32 = torch.float32
        pooled = self.buf("pooled", (nt, cc), f32)
        scores = self.buf("scores", (b, self.num_classes), f32)
        dscores = self.buf("dscores", (b, self.num_classes), f32)
        loss_part = self.buf("loss_part", (b,), f32)
        check(lib.mvf_head_train_fwd(_p(xcur), b, t, hc * wc, cc, _p(self.fc_w), _p(self.fc_b), self.num_classes, _p(lab), _p(mask), _p(pooled),
                                     _p(scores), _p(dscores), _p(loss_part), _p(loss), self.dt, _st()), "head fwd")
        self.saved.update(pooled=pooled, dscores=dscores, mask=mask, hw=hc * wc, c=cc, feat_shape=xcur.shape, scores=scores)
        if self._nbt_touched:
            self._count_batches()
        return loss

    def _count_batches(self):
        """num_batches_tracked += 1 of every BatchNorm in training mode: one launch (see _BN._count)."""
        key = tuple(m_.training for m_ in self._nbt_mods)
        if all(key):
            self._nbt_flat += 1
        else:                                 # BatchNorms in eval mode (norm_eval / frozen stages) do not count the step
            if key not in self._nbt_inc:
                self._nbt_inc[key] = torch.tensor([int(k) for k in key] or [0], dtype=torch.int64, device=self.device)
            self._nbt_flat += self._nbt_inc[key]
        self._nbt_touched = False

    def backward(self, exchange=False):
        """exchange=True (train_step): this engine also owns the data-parallel gradient exchange and may start it during
        backward; False (autograd API / external optimizer hooks): gradients are only produced."""
        self._main = torch.cuda.current_stream()
        self._exchange = bool(exchange)
        with _on_stream(self._main, main=True):
            self._backward()

    def _backward(self):
        s = self.saved
        if getattr(self, "_packs_on_side", False):
            self._wait(self.main_stream(), self._side)     # data-gradient weight packs (queued at the start of forward)
        nt, b, t = s["nt"], s["b"], s["t"]
        dpool = self.buf("dpool", (b, s["c"]), torch.float32)
        g = self.buf("gfeat", tuple(s["feat_shape"]))
        check(lib.mvf_head_train_bwd(_p(s["dscores"]), _p(s["pooled"]), _p(self.fc_w), _p(s["mask"]), b, t, s["hw"], s["c"], self.num_classes,
                                     _p(self.dfc_w), _p(self.dfc_b), _p(dpool), _p(g), self.dt, _st()), "head bwd")
        gated = sums = False
        for i in range(len(self.blocks) - 1, -1, -1):
            req = self.blocks[i - 1].wants_gated_gradient(self) if i > 0 else None       # the block below takes gm = g * [out > 0] as a tensor
            g = self.blocks[i].backward(g, nt, self, g_gated=gated, gate=req, sums_done=sums)
            gated, sums = self.blocks[i].gated_out, self.blocks[i].sums_out
            if i == self._tail_block:
                self._launch_tail_allreduce()
        ho, wo = s["ho"], s["wo"]
        # the stem's max-pool + ReLU + BatchNorm backward: a sums pass that gathers the pooled gradient through the argmax bytes (no scattered gradient tensor) and
        # produces the BatchNorm's backward sums, then an apply pass that gathers again and writes dz0 ([r3]: against scatter + reduce + apply: 515 -> 301 us)
        bn = self.stem_bn
        rows = lib.mvf_maxpool_bwd_sums_rows(nt, ho)
        part = self.buf("stem_bnsums", (64, rows, 2), torch.float32)
        check(lib.mvf_maxpool_bn_relu_bwd_sums(_p(s["amax"]), _p(g), nt, ho, wo, 64, None, _p(s["z0"]), _p(bn.mean), _p(bn.invstd),
                                               _p(bn.scale), _p(bn.shift), _p(part), self.dt, _st()), "maxpool bwd + bn sums")
        check(lib.mvf_bn_bwd_finalize(_p(part), rows, 64, _p(bn.dgamma), _p(bn.dbeta), _st()), "bn bwd finalize")
        dz0 = self.buf((id(bn), "dz"), s["z0"].shape, s["z0"].dtype)
        sg, sb = (bn._zero, bn._zero) if bn.frozen else (bn.dgamma, bn.dbeta)
        check(lib.mvf_maxpool_bn_relu_bwd_apply(_p(s["amax"]), _p(g), nt, ho, wo, 64, _p(s["z0"]), _p(bn.gamma), _p(bn.mean), _p(bn.invstd), _p(bn.scale),
                                                _p(bn.shift), _p(sg), _p(sb), _p(dz0), self.dt, _st()), "maxpool bwd + bn apply")
        # the step's last weight gradient runs on the LAUNCH stream: nothing is left there to overlap it with, and the side stream
        # still has layer1's weight gradients queued -- the two tails now run side by side
        self.flush_side()
        self.stem.wgrad(dz0, s["xp"], nt, s["hp"], s["wp"], ho, wo, self, on_main=True)
        self.join_side()
        if self.keep_io:
            self.io = dict(p0=self.buf("p0", (g.shape[0], 64)), g_p0=g, gfeat=self.buf("gfeat", tuple(s["feat_shape"])), nt=nt, b=b, t=t)
        self.saved = None

    exchange_enabled = True       # False: skip the gradient exchange (bench.py's "how much of the all-reduce is exposed" measurement only)

    def _ddp_active(self):
        import torch.distributed as dist
        return self.exchange_enabled and dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or self.force_allreduce)

    overlap_allreduce = _policy("ddp_overlap", 1) != 0      # the tail gradient bucket's all-reduce issued during backward

    def _launch_tail_allreduce(self):
        """Called from backward when layer3's first block is done: all-reduce flat_grads[tail_off:] (layer3, layer4, head: 94 % of
        the bytes) asynchronously while layer2 / layer1 / stem run their backward (about half of its time).  The collective is
        issued from the side stream after that stream has been ordered behind the main stream: at this point everything that
        writes the slice -- weight-gradient GEMMs and MVF tap gradients (side stream), BatchNorm / head gradients (main stream)
        -- is queued ahead of it, and nothing later writes it.  Every rank issues the same two collectives in the same order
        (tail here, head slice in allreduce_grads), which is all RCCL needs."""
        self._tail_launched = False
        if not (getattr(self, "_exchange", False) and self.overlap_allreduce and self._tail_off and self._ddp_active() and self.overlap_wgrad):
            return
        import torch.distributed as dist
        side = self.side_stream()
        self.flush_side()
        # The collective gets a stream of its own that is MEASURED to run beside both the launch stream and the weight-gradient
        # stream (streams.py: a stream that shares a hardware queue with either would serialise the ~1-2 ms RCCL kernel into it).
        # It is issued as a synchronous op with that stream current: ProcessGroupNCCL then either runs it on the current stream
        # (recent PyTorch) or on its internal stream with the current stream waiting for it -- in both cases `comm` is complete
        # when the collective is, and the optimizer orders itself behind `comm`.
        if getattr(self, "_comm", None) is None:
            from .streams import concurrent_stream
            self._comm = concurrent_stream(self.main_stream(), avoid=[side])

        def exchange():
            comm = self._comm
            comm.wait_stream(side)                 # weight-gradient GEMMs / tap gradients of layer3, layer4, head
            comm.wait_stream(self.main_stream())   # BatchNorm / head gradients
            with torch.cuda.stream(comm):
                dist.all_reduce(self.flat_grads[self._tail_off:])
            self._tail_launched = True
        if self._rec is not None:
            self._rec.py_op(exchange)              # a launch plan is cut here: the collective stays a torch call between its two segments
        else:
            exchange()

    def allreduce_grads(self):
        """reference dist_utils.py:38-49: the flat gradient is all-reduced (sum); the division by world size is folded into the
        optimizer kernel's grad_scale.  One collective, or two when the tail bucket was launched during backward."""
        import torch.distributed as dist
        if not self._ddp_active():
            return 1
        if self._tail_launched:
            # first order this stream behind the tail bucket's collective (long finished by now), THEN issue the second one: the
            # two collectives of the communicator never overlap, whatever stream ProcessGroupNCCL runs them on
            torch.cuda.current_stream().wait_stream(self._comm)
            dist.all_reduce(self.flat_grads[:self._tail_off])
            self._tail_launched = False
        else:
            dist.all_reduce(self.flat_grads)
        return dist.get_world_size()

    force_allreduce = False

    def step(self, lr=None):
        with _on_stream(torch.cuda.current_stream(), main=True):
            return self._step(lr)

    def _step(self, lr=None):
        world = self.allreduce_grads()
        if self.apply_sgd(lr, world) is None:
            return self.norm_out
        for m_ in (self.model.backbone, self.model.cls_head):
            if hasattr(m_, "invalidate_engine"):
                m_.invalidate_engine()
        return self.norm_out

    # [r6] launch plans (launch_plan.py): forward + backward of train_step recorded once per (batch shape, engine switches) and replayed from C
    # (mvf_plan_run); MVF_POLICY=plan=0 keeps every step on the Python launch sequence
    use_plan = _policy("plan", 1) != 0
    plan_warmup = 2          # eager steps before the first recording (buffers and workspaces reach their final sizes in the first two)

    def _plan_key(self, imgs, labels):
        """None = this step cannot run from a plan; else what a plan is valid for."""
        if not (self.use_plan and imgs.dtype == torch.float32 and imgs.is_contiguous() and not self.keep_io and self.input_pipeline is None):
            return None
        if not all(m_.training for m_ in self._nbt_mods):
            return None                  # frozen statistics are folded per step by torch calls (_BN.use_running_stats)
        sw = self.__dict__.get("_switch_names")
        if sw is None:
            sw = self._switch_names = sorted(k for c_ in type(self).__mro__ for k, v in vars(c_).items()
                                             if not k.startswith("_") and isinstance(v, (bool, int, float, str)) and k not in ("lr", "momentum", "weight_decay", "max_norm"))
        return (tuple(imgs.shape), tuple(labels.shape), labels.dtype, str(imgs.device), torch.cuda.current_stream().cuda_stream, self._ddp_active(),
                self.dropout > 0.0, tuple(p.requires_grad for p in self.model.parameters()), tuple(getattr(self, k) for k in sw))

    def _record_step(self, imgs, labels, prepared):
        """One eager step with every library call and stream ordering recorded; returns (loss, Plan | None)."""
        from . import launch_plan
        import sys
        mod = sys.modules[__name__]
        rec = launch_plan.Recorder()
        allocs0 = torch.cuda.memory_stats(imgs.device).get("allocation.all.allocated", 0)
        real_lib, self._rec = mod.lib, rec
        mod.lib = launch_plan.RecordingLib(rec)
        try:
            loss = self.forward(imgs, labels, _prepared=prepared)
            self.backward(exchange=True)
        finally:
            mod.lib, self._rec = real_lib, None
        # a tensor allocated INSIDE the step would be freed after it, and a replay would launch on memory the allocator has handed to somebody else
        if torch.cuda.memory_stats(imgs.device).get("allocation.all.allocated", 0) != allocs0:
            rec.unsupported = "the step allocates device memory"
        if rec.unsupported:
            return loss, None
        lab, mask, _ = prepared
        plan = launch_plan.Plan(rec, dict(imgs=imgs.data_ptr(), labels=lab.data_ptr(), mask=mask.data_ptr() if mask is not None else 0, loss=loss.data_ptr()))
        plan.storage_epoch = self._storage_epoch
        return loss, plan

    def train_step(self, imgs, labels, lr=None):
        key = self._plan_key(imgs, labels) if imgs.is_cuda else None
        if key is None:
            loss = self.forward(imgs, labels)
            self.backward(exchange=True)
            self.step(lr)
            return loss
        st = self.__dict__.setdefault("_plans", {}).setdefault(key, dict(eager=0, tries=0, cand=None, plan=None))
        prepared = self._step_tensors(imgs, labels)
        plan = st["plan"]
        if plan is not None and plan.storage_epoch != self._storage_epoch:
            # a larger batch (or workspace) replaced storage this plan points into since it was recorded: drop it and record this shape again
            plan = st["plan"] = st["cand"] = None
            st["tries"] = 0
        if plan is not None:
            lab, mask, loss = prepared
            self._main = torch.cuda.current_stream()
            self.forward_count += 1
            self._exchange = True
            self._count_batches()
            plan.run(dict(imgs=imgs.data_ptr(), labels=lab.data_ptr(), mask=mask.data_ptr() if mask is not None else 0, loss=loss.data_ptr()))
        elif st["eager"] < self.plan_warmup or st["tries"] >= 4:
            st["eager"] += 1
            loss = self.forward(imgs, labels, _prepared=prepared)
            self.backward(exchange=True)
        else:
            st["tries"] += 1
            loss, cand = self._record_step(imgs, labels, prepared)
            if cand is not None and st["cand"] is not None and cand.signature == st["cand"].signature and cand.storage_epoch == self._storage_epoch:
                st["plan"], st["cand"] = cand, None          # two consecutive steps made the same calls with the same arguments
            else:
                st["cand"] = cand
        self.step(lr)
        return loss
