"""Inference engine: the whole MVFNet backbone + head as a fixed sequence of HIP launches (C ABI).

Host-side mirror of ResNet.forward (reference codes/models/backbones/resnet.py:479-494), Bottleneck.forward
(:208-244), MVF.forward (codes/models/modules/MVF.py:104-138) and TSNClsHead.forward
(codes/models/heads/tsn_clshead.py:71-117) for eval-mode BatchNorm:

  * activations are channels-last (N*T, H, W, C) end to end; the only layout change is the stem's
    NCHW -> padded NHWC4 re-pack of the 3-channel input;
  * every BatchNorm2d is folded into the packed conv weights (scale) and an epilogue bias (shift); ReLU and
    the residual add run in the conv epilogue -- no elementwise kernels at all;
  * MVF writes only its C/8 slice to a compact buffer; the wrapped 1x1 conv reads channels [0,Cs) from that
    buffer and [Cs,C) from the block input (split-A operand), so the reference's cat/transpose/contiguous
    copies of the whole tensor disappear and the block input stays intact for the residual;
  * PyTorch supplies device memory (caching allocator) and the current stream, nothing else.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ConvDesc, MvfDesc, check, lib

_DT = {torch.float32: _lib.MVF_F32, torch.bfloat16: _lib.MVF_BF16}


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_SK_WS = {}
# MVF computed INSIDE conv1's A-operand loader (mvf_conv2d_nhwc_fwd_mvf: one launch, no slice buffer) instead of the stencil kernel +
# split-A conv.  Parity-tested (tests/test_conv_gpu.py, tests/test_net_gpu.py), but measured 3-5 % SLOWER end to end on MI355X (bf16
# inference 7090-7270 vs 7480 clips/s, fp32 1622-1656 vs 1672): every n-tile of the conv recomputes the stencil of its rows (2-4x
# redundant VALU + loads) and the loader's extra registers cost a workgroup per CU -- so it is opt-in (MVF_POLICY=fuse_loader=1).
from .policy import policy as _policy      # noqa: E402
FUSE_MVF_LOADER = _policy("fuse_loader", 0) == 1


def _sk_workspace(device):
    """Stream-K scratch (partial accumulators + flags) for the conv kernel: one buffer per (device, stream)."""
    key = (str(device), torch.cuda.current_stream().cuda_stream)
    ws = _SK_WS.get(key)
    if ws is None:
        ws = torch.empty(lib.mvf_conv2d_workspace_bytes(None), dtype=torch.uint8, device=device)
        _SK_WS[key] = ws
    return ws


class _Conv(object):
    """One packed conv (+ folded BN) and its launch descriptor."""

    def __init__(self, conv, bn, dtype, relu, device, stem=False):
        w = conv.weight.detach().to(device=device, dtype=torch.float32).contiguous()
        cout, cin, kh, kw = w.shape
        self.dt = dtype
        self.relu = int(relu)
        self.cout, self.stride, self.pad = cout, conv.stride[0], conv.padding[0]
        st = _stream()
        scale = shift = None
        if bn is not None:
            f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()
            scale = torch.empty(cout, dtype=torch.float32, device=device)
            shift = torch.empty(cout, dtype=torch.float32, device=device)
            check(lib.mvf_bn_fold(_p(f32(bn.weight)), _p(f32(bn.bias)), _p(f32(bn.running_mean)), _p(f32(bn.running_var)),
                                  C.c_float(bn.eps), cout, _p(scale), _p(shift), st), "mvf_bn_fold")
        self.bias = shift
        if stem:
            # 7x7/2 stem on the padded NHWC4 input: one K chunk per kernel row = 8 pixels x 4 channels
            if cin > 4 or kw > 8:
                raise NotImplementedError("stem packing supports <= 4 input channels and kernel width <= 8")
            self.kh, self.kw, self.cin = kh, 1, 32
            self.wp = torch.empty(cout, kh, 8, 4, dtype=dtype, device=device)
            check(lib.mvf_pack_conv_weight(_p(w), cout, cin, kh, kw, 8, 4, _p(scale), _p(self.wp), _DT[dtype], st), "mvf_pack_conv_weight")
            self.stem_pad = conv.padding[0]
        else:
            ue = 4 if dtype == torch.float32 else 8
            if cin % ue:
                raise NotImplementedError("conv with cin=%d: channels must be a multiple of %d" % (cin, ue))
            self.kh, self.kw, self.cin = kh, kw, cin
            self.wp = torch.empty(cout, kh, kw, cin, dtype=dtype, device=device)
            check(lib.mvf_pack_conv_weight(_p(w), cout, cin, kh, kw, kw, cin, _p(scale), _p(self.wp), _DT[dtype], st), "mvf_pack_conv_weight")

    def out_hw(self, h, w):
        return (h + 2 * self.pad - self.kh) // self.stride + 1, (w + 2 * self.pad - self.kw) // self.stride + 1

    def run(self, x, n, h, w, c_total, residual=None, x2=None, split_c=0, ho=None, wo=None, stride=None, pad=None, out=None):
        stride = self.stride if stride is None else stride
        pad = self.pad if pad is None else pad
        if ho is None:
            ho, wo = self.out_hw(h, w)
        y = out if out is not None else torch.empty(n, ho, wo, self.cout, dtype=self.dt, device=x.device)
        if tuple(y.shape) != (n, ho, wo, self.cout) or y.dtype != self.dt or not y.is_contiguous():
            raise ValueError("conv output buffer %s does not match (%d, %d, %d, %d)" % (tuple(y.shape), n, ho, wo, self.cout))
        d = ConvDesc(n, h, w, self.cin, self.cout, self.kh, self.kw, stride, pad, ho, wo, c_total, _DT[self.dt], self.relu,
                     split_c, split_c)
        ws = _sk_workspace(x.device)
        check(lib.mvf_conv2d_nhwc_fwd_ws(C.byref(d), _p(x), _p(x2), _p(self.wp), _p(self.bias), _p(residual), _p(y), _p(ws), ws.numel(),
                                         _stream()), "mvf_conv2d_nhwc_fwd")
        return y, ho, wo

    def run_mvf(self, x, n, h, w, mvf, out=None):
        """conv1 with the MVF module fused into its A-operand load (one launch: no slice buffer, no stencil kernel)."""
        y = out if out is not None else torch.empty(n, h, w, self.cout, dtype=self.dt, device=x.device)
        d = ConvDesc(n, h, w, self.cin, self.cout, 1, 1, 1, 0, h, w, self.cin, _DT[self.dt], 1, 0, 0)
        ws = _sk_workspace(x.device)
        check(lib.mvf_conv2d_nhwc_fwd_mvf(C.byref(d), _p(x), _p(self.wp), _p(self.bias), _p(mvf.coef), mvf.cs, mvf.T, mvf.act, _p(y), _p(ws),
                                          ws.numel(), _stream()), "mvf_conv2d_nhwc_fwd_mvf")
        return y, h, w


class _MvfStage(object):
    def __init__(self, mvf, dtype, device):
        cs = mvf.num_shift_channel
        self.cs, self.T = cs, mvf.n_segment
        self.mode = _lib.MODE_BITS[mvf.mode]
        f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()
        self.wt = f32(mvf.shift_conv.weight).reshape(cs, 3)
        self.wh = self.wt if mvf.share else (f32(mvf.h_conv.weight).reshape(cs, 3) if self.mode & 2 else None)
        self.ww = self.wt if mvf.share else (f32(mvf.w_conv.weight).reshape(cs, 3) if self.mode & 4 else None)
        self.scale = self.shift = None
        if mvf.use_hs:
            bn = mvf.bn
            self.scale = torch.empty(cs, dtype=torch.float32, device=device)
            self.shift = torch.empty(cs, dtype=torch.float32, device=device)
            check(lib.mvf_bn_fold(_p(f32(bn.weight)), _p(f32(bn.bias)), _p(f32(bn.running_mean)), _p(f32(bn.running_var)),
                                  C.c_float(bn.eps), cs, _p(self.scale), _p(self.shift), _stream()), "mvf_bn_fold")
        self.dt = dtype
        # coefficient table of the FUSED path (mvf_conv2d_nhwc_fwd_mvf): [cs][12] = {w_t[3], w_h[3], w_w[3], scale, shift, 0}
        z3 = torch.zeros(cs, 3, dtype=torch.float32, device=device)
        one, zero = torch.ones(cs, 1, dtype=torch.float32, device=device), torch.zeros(cs, 1, dtype=torch.float32, device=device)
        self.act = int(mvf.use_hs)
        self.coef = torch.cat([self.wt, self.wh if self.wh is not None else z3, self.ww if self.ww is not None else z3,
                               self.scale.view(cs, 1) if self.scale is not None else one,
                               self.shift.view(cs, 1) if self.shift is not None else zero, zero], dim=1).contiguous()

    def run_slice(self, x, nt, h, w, c):
        out = torch.empty(nt, h, w, self.cs, dtype=self.dt, device=x.device)
        d = MvfDesc(nt, c, h, w, self.T, self.cs, self.mode, _lib.MVF_NHWC, _DT[self.dt])
        check(lib.mvf_fwd_infer_slice(C.byref(d), _p(x), _p(out), _p(self.wt), _p(self.wh), _p(self.ww), _p(self.scale),
                                      _p(self.shift), _stream()), "mvf_fwd_infer_slice")
        return out

    def run_full(self, x, nt, h, w, c):
        out = torch.empty(nt, h, w, c, dtype=self.dt, device=x.device)
        d = MvfDesc(nt, c, h, w, self.T, self.cs, self.mode, _lib.MVF_NHWC, _DT[self.dt])
        check(lib.mvf_fwd_infer(C.byref(d), _p(x), _p(out), _p(self.wt), _p(self.wh), _p(self.ww), _p(self.scale),
                                _p(self.shift), _stream()), "mvf_fwd_infer")
        return out


class _Block(object):
    def __init__(self, blk, dtype, device):
        from .modules.MVF import MVF
        c1 = blk.conv1
        self.mvf = None
        if isinstance(c1, MVF):
            if c1.num_shift_channel:
                self.mvf = _MvfStage(c1, dtype, device)
            c1 = c1.net
        self.c1 = _Conv(c1, blk.bn1, dtype, True, device)
        self.c2 = _Conv(blk.conv2, blk.bn2, dtype, True, device)
        self.c3 = _Conv(blk.conv3, blk.bn3, dtype, True, device)      # ReLU after the residual add
        self.down = None
        if blk.downsample is not None:
            self.down = _Conv(blk.downsample[0], blk.downsample[1], dtype, False, device)
        ce = 32 if dtype == torch.float32 else 64
        self.split_ok = self.mvf is not None and self.mvf.cs % ce == 0
        # MVF fused into conv1's loader (opt-in, see FUSE_MVF_LOADER); default: stencil kernel into a slice buffer + split-A conv
        self.fuse_mvf = self.split_ok and FUSE_MVF_LOADER and self.c1.kh == 1 and self.c1.stride == 1 and self.c1.relu

    def run(self, x, nt, h, w, c, out=None):
        if self.mvf is None:
            o1, _, _ = self.c1.run(x, nt, h, w, c)
        elif self.fuse_mvf:
            o1, _, _ = self.c1.run_mvf(x, nt, h, w, self.mvf)
        elif self.split_ok:
            sl = self.mvf.run_slice(x, nt, h, w, c)
            o1, _, _ = self.c1.run(x, nt, h, w, c, x2=sl, split_c=self.mvf.cs)
        else:
            o1, _, _ = self.c1.run(self.mvf.run_full(x, nt, h, w, c), nt, h, w, c)
        o2, ho, wo = self.c2.run(o1, nt, h, w, self.c1.cout)
        idn = x
        if self.down is not None:
            idn, _, _ = self.down.run(x, nt, h, w, c)
        o3, _, _ = self.c3.run(o2, nt, ho, wo, self.c2.cout, residual=idn, out=out)
        return o3, ho, wo, self.c3.cout


class BackboneEngine(object):
    """Packed, BN-folded copy of a mvfnet_amd ResNet for eval-mode inference on one GPU."""

    def __init__(self, resnet, dtype=torch.float32, device=None):
        if dtype not in _DT:
            raise TypeError("engine dtype must be float32 or bfloat16")
        device = device or next(resnet.parameters()).device
        if torch.device(device).type != "cuda":
            raise RuntimeError("BackboneEngine needs the model on an MI355X device (got %s); no CPU fallback" % device)
        self.dtype, self.device = dtype, device
        self.input_pipeline, self.input_window = None, None     # uint8 frame input (preprocess.FramePipeline), see forward()
        with torch.no_grad():
            self.stem = _Conv(resnet.conv1, resnet.bn1, dtype, True, device, stem=True)
            self.blocks = []
            for name in resnet.res_layers:
                for blk in getattr(resnet, name):
                    self.blocks.append(_Block(blk, dtype, device))
        self.stage_ends = []
        k = 0
        for name in resnet.res_layers:
            k += len(getattr(resnet, name))
            self.stage_ends.append(k)

    def forward(self, x_nchw, stages=None, n_segment=None):
        """x_nchw: (NT, 3, H, W) fp32 contiguous on the GPU -> features (NT, h, w, 2048) channels-last buffer.

        With `self.streams` > 1 (and no stage capture) the batch is cut into that many groups of whole clips, each
        run as an independent launch chain on its own HIP stream: clips are independent units in eval mode, and
        two chains in flight let one chain's kernels fill the CUs another chain's last (partial) wave of tiles
        leaves idle."""
        u8 = x_nchw.dtype == torch.uint8             # decoded (NT, Hs, Ws, 3) frames + self.input_pipeline / self.input_window
        if not x_nchw.is_cuda or not (x_nchw.dtype == torch.float32 or u8):
            raise RuntimeError("engine input must be a float32 GPU tensor (got %s on %s)" % (x_nchw.dtype, x_nchw.device))
        x_nchw = x_nchw.contiguous()
        window = getattr(self, "input_window", None) if u8 else None
        ns = getattr(self, "streams", 1)
        T = n_segment or self._n_segment()
        clips = x_nchw.shape[0] // max(T, 1)
        if ns > 1 and stages is None and clips >= 2 * ns and x_nchw.shape[0] % T == 0:
            return self._forward_multi(x_nchw, ns, T, window)
        return self._forward_one(x_nchw, stages, window)

    def _n_segment(self):
        for b in self.blocks:
            if b.mvf is not None:
                return b.mvf.T
        return 1

    def feature_shape(self, h, w):
        """(h, w, c) of the features for an h x w input (the shape arithmetic of _forward_one, no kernels)."""
        pad = self.stem.stem_pad
        ho, wo = (h + 2 * pad - self.stem.kh) // 2 + 1, (w + 2 * pad - 7) // 2 + 1
        h, w = (ho + 2 - 3) // 2 + 1, (wo + 2 - 3) // 2 + 1
        for blk in self.blocks:
            h, w = blk.c2.out_hw(h, w)
        return h, w, self.blocks[-1].c3.cout

    def _forward_multi(self, x, ns, T, window=None):
        cur = torch.cuda.current_stream()
        if not hasattr(self, "_pool") or len(self._pool) != ns:
            from .streams import concurrent_stream
            self._pool = []
            for _ in range(ns):                      # chains on streams that share a hardware queue would just run in turn
                self._pool.append(concurrent_stream(cur, avoid=self._pool))
        clips = x.shape[0] // T
        per = (clips + ns - 1) // ns
        # every chain's last conv writes its clips straight into one feature tensor (allocated on the caller's stream): no
        # concatenation copy afterwards
        if x.dtype == torch.uint8:
            hin, win = self.input_pipeline.crop_hw
        else:
            hin, win = x.shape[2], x.shape[3]
        hf, wf, cf = self.feature_shape(hin, win)
        feat = torch.empty(x.shape[0], hf, wf, cf, dtype=self.dtype, device=x.device)
        used = []
        for i, st in enumerate(self._pool):
            lo, hi = i * per * T, min((i + 1) * per, clips) * T
            if lo >= hi:
                continue
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                self._forward_one(x[lo:hi], None, None if window is None else window.reshape(-1, 3)[lo:hi], out=feat[lo:hi])
            used.append(st)
        for st in used:
            cur.wait_stream(st)
        return feat

    def _forward_one(self, x_nchw, stages=None, window=None, out=None):
        pad = self.stem.stem_pad
        if x_nchw.dtype == torch.uint8:              # decoded (nt, Hs, Ws, 3) frames: crop / flip / normalise fused into the stem prep
            if self.input_pipeline is None:
                raise RuntimeError("uint8 frames need engine.input_pipeline = preprocess.FramePipeline(...)")
            nt = x_nchw.shape[0]
            h, w = self.input_pipeline.crop_hw
            hp, wp = h + 2 * pad, (w + 2 * pad + 2 + 1) // 2 * 2
            xp = self.input_pipeline.to_stem(x_nchw, window, pad, wp, self.dtype)
        else:
            nt, cin, h, w = x_nchw.shape
            hp, wp = h + 2 * pad, (w + 2 * pad + 2 + 1) // 2 * 2
            xp = torch.empty(nt, hp, wp, 4, dtype=self.dtype, device=x_nchw.device)
            check(lib.mvf_stem_prep(_p(x_nchw), nt, cin, h, w, pad, wp, _p(xp), _DT[self.dtype], _stream()), "mvf_stem_prep")
        ho, wo = (h + 2 * pad - self.stem.kh) // 2 + 1, (w + 2 * pad - 7) // 2 + 1
        y, _, _ = self.stem.run(xp, nt, hp, wp, 4, ho=ho, wo=wo, stride=2, pad=0)
        h2, w2 = (ho + 2 - 3) // 2 + 1, (wo + 2 - 3) // 2 + 1
        p = torch.empty(nt, h2, w2, self.stem.cout, dtype=self.dtype, device=x_nchw.device)
        check(lib.mvf_maxpool3x3s2_nhwc(_p(y), nt, ho, wo, self.stem.cout, _p(p), _DT[self.dtype], _stream()), "mvf_maxpool3x3s2_nhwc")
        if stages is not None:
            stages["maxpool"] = p
        x, h, w, c = p, h2, w2, self.stem.cout
        for i, blk in enumerate(self.blocks):
            x, h, w, c = blk.run(x, nt, h, w, c, out=out if i == len(self.blocks) - 1 else None)
            if stages is not None and (i + 1) in self.stage_ends:
                stages["layer%d" % (self.stage_ends.index(i + 1) + 1)] = x
        return x


class HeadEngine(object):
    """TSN head + clip averaging (tsn_clshead.py:71-117, base.py:43-74) on the channels-last features."""

    KINDS = {None: 0, "score": 1, "prob": 2}

    def __init__(self, fc, device):
        """fc: the head's nn.Linear.  Its parameters are read at EVERY call (fp32 contiguous GPU parameters are used in place, no
        copy): a training engine re-points them at its flat buffer and updates them in place, so a cached copy would score with
        stale weights after the first optimizer step."""
        self.fc, self.device = fc, device
        self.classes, self.c = fc.weight.shape

    @property
    def w(self):
        return self.fc.weight.detach().to(device=self.device, dtype=torch.float32).contiguous()

    @property
    def b(self):
        return self.fc.bias.detach().to(device=self.device, dtype=torch.float32).contiguous() if self.fc.bias is not None else None

    def scores(self, feat, num_seg):
        nt, h, w, c = feat.shape
        if c != self.c or nt % num_seg:
            raise ValueError("head: features %s do not match in_channels=%d / num_seg=%d" % (tuple(feat.shape), self.c, num_seg))
        clips = nt // num_seg
        pooled = torch.empty(clips, c, dtype=torch.float32, device=feat.device)
        out = torch.empty(clips, self.classes, dtype=torch.float32, device=feat.device)
        wgt, bias = self.w, self.b            # kept alive until the launch is queued
        check(lib.mvf_head_pool_fc(_p(feat), clips, num_seg, h * w, c, _p(wgt), _p(bias), self.classes, _p(pooled), _p(out),
                                   _DT[feat.dtype], _stream()), "mvf_head_pool_fc")
        return out

    def average(self, scores, average_clips):
        if average_clips not in self.KINDS:
            raise ValueError("%s is not supported. Currently supported ones are ['score', 'prob', None]" % (average_clips,))
        kind = self.KINDS[average_clips]
        if kind == 0:
            return scores
        clips, classes = scores.shape
        out = torch.empty(1, classes, dtype=torch.float32, device=scores.device)
        check(lib.mvf_average_clip(_p(scores), clips, classes, kind, _p(out), _stream()), "mvf_average_clip")
        return out
