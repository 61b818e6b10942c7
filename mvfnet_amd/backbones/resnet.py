"""2D ResNet-50/101/152 backbone, MI355X-native.

Mirror of the reference's `codes/models/backbones/resnet.py` (Bottleneck :104-244, make_res_layer :247-326,
ResNet :329-527) for the configurations MVFNet ships (style='pytorch', no dilation, plain stem): identical
constructor arguments, sub-module names and state_dict keys (conv1/bn1/layerL.B.{conv1,bn1,conv2,bn2,conv3,bn3,
downsample.0,downsample.1}), so reference checkpoints load as they are.

The nn.Conv2d / nn.BatchNorm2d children are PARAMETER CONTAINERS: forward() never calls them.  In eval mode
the whole stack runs through mvfnet_amd.engine.BackboneEngine (hand-written HIP kernels behind the C ABI).
"""
import torch
import torch.nn as nn
from torch.nn.modules.batchnorm import _BatchNorm

from ..builder import BACKBONES

_NORMS = {"BN": nn.BatchNorm2d}


def _norm(norm_cfg, n):
    cfg = dict(norm_cfg or dict(type="BN"))
    kind = cfg.pop("type")
    if kind not in _NORMS:
        raise KeyError("Unrecognized / unsupported norm type %s (MVFNet configs use 'BN')" % kind)
    requires_grad = cfg.pop("requires_grad", True)
    cfg.setdefault("eps", 1e-5)                         # reference: codes/models/common/norm.py:59
    layer = _NORMS[kind](n, **cfg)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return layer


class Bottleneck(nn.Module):
    """conv1(1x1) - bn1 - relu - conv2(3x3, stride) - bn2 - relu - conv3(1x1) - bn3 - (+identity) - relu."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style="pytorch",
                 norm_cfg=dict(type="BN"), with_cp=False, avd=False, avd_first=False):
        super().__init__()
        # with_cp (activation checkpointing, reference resnet.py:237-240) trades recomputation for memory and changes no result: the
        # engine keeps every activation of a step resident (a C3 step holds ~25 GB of the 288 GB), so the flag is accepted and ignored
        if style != "pytorch" or dilation != 1 or avd:
            raise NotImplementedError("Bottleneck: only style='pytorch', dilation=1, avd=False are built "
                                      "(the options the MVFNet configs use)")
        self.inplanes, self.planes = inplanes, planes
        self.conv1_stride, self.conv2_stride = 1, stride
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, stride=1, bias=False)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = _norm(norm_cfg, planes)
        self.bn2 = _norm(norm_cfg, planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = _norm(norm_cfg, planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride, self.dilation, self.norm_cfg, self.with_cp = stride, dilation, norm_cfg, with_cp

    norm1 = property(lambda self: self.bn1)
    norm2 = property(lambda self: self.bn2)
    norm3 = property(lambda self: self.bn3)

    def forward(self, x):
        """Stand-alone block (reference resnet.py:208-244): (N*T, C, H, W) CUDA tensor -> (N*T, 4 * planes, H/s, W/s) through the HIP block
        kernels (`train_engine.BlockTrainer`: conv + BatchNorm (batch statistics when the BatchNorm is in training mode, folded running
        statistics in eval mode) + ReLU + MVF + residual), as ONE autograd node whose backward is the HIP backward of the block.  Inside a
        ResNet / Recognizer2D the blocks are not called one by one: the whole stack runs as one fused launch sequence (ResNet.engine(),
        Recognizer2D.forward_train).  Storage type = x.dtype (float32, or bfloat16 storage with fp32 accumulation / statistics / gradients)."""
        if not x.is_cuda:
            raise RuntimeError("mvfnet_amd Bottleneck runs on MI355X tensors only; no CPU fallback (tests use oracle/)")
        if x.dtype not in (torch.float32, torch.bfloat16):
            raise TypeError("Bottleneck.forward: float32 or bfloat16 input, got %s" % x.dtype)
        tr = getattr(self, "_trainer", None)
        if tr is None or tr.tdtype != x.dtype:
            from ..train_engine import BlockTrainer
            # rehome=False: the block's parameters stay where they are (they may be views of a model-level TrainEngine's flat buffer,
            # which the optimizer keeps updating); only the gradient buffer is the trainer's own, and backward() hands out copies
            tr = BlockTrainer(self, dtype=x.dtype, rehome=False)
            object.__setattr__(self, "_trainer", tr)
        return _BlockFn.apply(tr, x, *list(self.parameters()))

    def invalidate_engine(self):
        object.__setattr__(self, "_trainer", None)

    def _apply(self, fn, *a, **k):
        object.__setattr__(self, "_trainer", None)      # the trainer's flat buffers alias the OLD parameter storage
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        object.__setattr__(self, "_trainer", None)
        return super().load_state_dict(*a, **k)


class _BlockFn(torch.autograd.Function):
    """One bottleneck as an autograd node: forward / backward = the HIP block kernels (BlockTrainer keeps ONE forward's activations)."""

    @staticmethod
    def forward(ctx, tr, x, *params):
        ctx.tr = tr
        y = tr.forward(x.detach())
        tr.forward_token = getattr(tr, "forward_token", 0) + 1
        ctx.token = tr.forward_token
        return y.to(x.dtype)

    @staticmethod
    def backward(ctx, gy):
        tr = ctx.tr
        if ctx.token != getattr(tr, "forward_token", 0):
            raise RuntimeError("Bottleneck: backward() of an output whose activations are gone (the block keeps one forward)")
        tr.forward_token += 1                               # one backward per forward: the block's activations are consumed
        dx = tr.backward(gy.contiguous())
        grads = []
        for p in tr.model.parameters():
            grads.append(tr.grad_of(p).clone() if p.requires_grad else None)
        return (None, dx.to(gy.dtype)) + tuple(grads)


def make_res_layer(block, inplanes, planes, blocks, stride=1, dilation=1, style="pytorch", norm_cfg=None,
                   with_cp=False, avg_down=False, avd=False, avd_first=False):
    if avg_down:
        raise NotImplementedError("avg_down is not used by MVFNet configs")
    downsample = None
    if stride != 1 or inplanes != planes * block.expansion:
        downsample = nn.Sequential(
            nn.Conv2d(inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
            _norm(norm_cfg, planes * block.expansion))
    layers = [block(inplanes, planes, stride, dilation, downsample, style=style, norm_cfg=norm_cfg, with_cp=with_cp)]
    for _ in range(1, blocks):
        layers.append(block(planes * block.expansion, planes, 1, dilation, style=style, norm_cfg=norm_cfg, with_cp=with_cp))
    return nn.Sequential(*layers)


@BACKBONES.register_module
class ResNet(nn.Module):
    arch_settings = {50: (Bottleneck, (3, 4, 6, 3)), 101: (Bottleneck, (3, 4, 23, 3)), 152: (Bottleneck, (3, 8, 36, 3))}

    def __init__(self, depth, pretrained=None, in_channels=3, num_stages=4, strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1),
                 out_indices=(0, 1, 2, 3), style="pytorch", frozen_stages=-1, norm_cfg=dict(type="BN", requires_grad=True),
                 norm_eval=True, norm_frozen=False, partial_norm=False, with_cp=False, avg_down=False, avd=False,
                 avd_first=False, deep_stem=False, stem_width=64, engine_dtype=torch.float32):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError("invalid depth %s for resnet (bottleneck depths 50/101/152 are built)" % depth)
        if deep_stem or avg_down or avd or tuple(dilations) != (1,) * len(dilations) or tuple(strides)[:num_stages] != (1, 2, 2, 2)[:num_stages]:
            raise NotImplementedError("ResNet: only the plain stem / stride (1,2,2,2) / dilation 1 variant is built")
        if not (1 <= num_stages <= 4) or max(out_indices) >= num_stages:
            raise ValueError("bad num_stages / out_indices")
        self.depth, self.in_channels, self.pretrained = depth, in_channels, pretrained
        self.num_stages, self.strides, self.dilations = num_stages, strides, dilations
        self.out_indices, self.style, self.frozen_stages = out_indices, style, frozen_stages
        self.norm_cfg, self.norm_eval, self.norm_frozen, self.partial_norm = norm_cfg, norm_eval, norm_frozen, partial_norm
        self.block, stage_blocks = self.arch_settings[depth]
        self.stage_blocks = stage_blocks[:num_stages]
        self.inplanes = 64
        self.conv1 = nn.Conv2d(in_channels, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = _norm(norm_cfg, 64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.res_layers = []
        for i, nblk in enumerate(self.stage_blocks):
            planes = 64 * 2 ** i
            layer = make_res_layer(self.block, self.inplanes, planes, nblk, stride=strides[i], dilation=1, style=style,
                                   norm_cfg=norm_cfg)
            self.inplanes = planes * self.block.expansion
            name = "layer%d" % (i + 1)
            self.add_module(name, layer)
            self.res_layers.append(name)
        self.feat_dim = self.block.expansion * 64 * 2 ** (len(self.stage_blocks) - 1)
        self.engine_dtype = engine_dtype
        self._engine = None

    norm1 = property(lambda self: self.bn1)

    # ---- weights -----------------------------------------------------------------------------------------
    def init_weights(self):
        """reference resnet.py:464-477: kaiming-normal(fan_out, relu) convs, BN weight 1 / bias 0; a checkpoint
        path is loaded non-strictly (`module.` prefixes stripped)."""
        if isinstance(self.pretrained, str):
            from ..checkpoint import load_checkpoint
            load_checkpoint(self, self.pretrained, strict=False)
        elif self.pretrained is None:
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                elif isinstance(m, nn.BatchNorm2d):
                    nn.init.constant_(m.weight, 1)
                    nn.init.constant_(m.bias, 0)
        else:
            raise TypeError("pretrained must be a str or None")
        self.invalidate_engine()

    # ---- engine plumbing ----------------------------------------------------------------------------------
    def invalidate_engine(self):
        self._engine = None

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    def engine(self):
        if self._engine is None:
            from ..engine import BackboneEngine
            self._engine = BackboneEngine(self, self.engine_dtype)
        self._engine.input_pipeline = getattr(self, "input_pipeline", None)      # uint8 frame input (preprocess.FramePipeline)
        return self._engine

    def _bn_all_eval(self):
        return not any(m.training for m in self.modules() if isinstance(m, _BatchNorm))

    def forward(self, x, stages=None):
        """(N*T, 3, H, W) -> (N*T, 2048, h, w) features (a channels-last strided view of the engine's NHWC buffer).
        Mirrors ResNet.forward with out_indices=(3,) (reference resnet.py:479-494)."""
        if tuple(self.out_indices) != (self.num_stages - 1,):
            raise NotImplementedError("only out_indices=(last stage,) is built (MVFNet configs use (3,))")
        if not self._bn_all_eval():
            raise RuntimeError(
                "mvfnet_amd ResNet.forward is the eval-mode (folded BatchNorm) inference path. With BatchNorms in training mode the "
                "stack runs -- forward with batch statistics AND backward -- inside Recognizer2D.forward_train / "
                "mvfnet_amd.train_engine.TrainEngine (one fused launch sequence for backbone + head + loss); call .eval() here")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and x.requires_grad:
            raise RuntimeError("mvfnet_amd ResNet.forward does not record an autograd graph (inference path); gradients come from "
                               "Recognizer2D.forward_train(...)['loss_cls'].backward() -- wrap this call in torch.no_grad()")
        feat = self.engine().forward(x, stages)
        return feat.permute(0, 3, 1, 2)

    def train(self, mode=True):
        """reference resnet.py:496-527: norm_eval keeps every BN in eval mode (optionally frozen); frozen_stages
        freezes the stem and the first stages."""
        super().train(mode)
        self._engine = None
        if self.norm_eval:
            for m in self.modules():
                if isinstance(m, _BatchNorm):
                    m.eval()
                    if self.norm_frozen:
                        for p in m.parameters():
                            p.requires_grad = False
        if self.partial_norm:
            for i in range(1, self.frozen_stages + 1):
                for m in getattr(self, "layer%d" % i).modules():
                    if isinstance(m, _BatchNorm):
                        m.eval()
                        m.weight.requires_grad = False
                        m.bias.requires_grad = False
        if mode and self.frozen_stages >= 0:
            for p in list(self.conv1.parameters()) + list(self.bn1.parameters()):
                p.requires_grad = False
            self.bn1.eval()
            for i in range(1, self.frozen_stages + 1):
                mod = getattr(self, "layer%d" % i)
                mod.eval()
                for p in mod.parameters():
                    p.requires_grad = False
        return self
