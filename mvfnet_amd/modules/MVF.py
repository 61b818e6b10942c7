"""MVF -- Multi-View Fusion module, MI355X-native.

Drop-in for the reference's `codes/models/modules/MVF.py` (class MVF :53-138, make_multi_view_fusion :18-49):
same constructor signature, same attribute and sub-module names, same state_dict keys/shapes
(`net, shift_conv (Cs,1,3,1,1), h_conv (Cs,1,1,3,1), w_conv (Cs,1,1,1,3), bn.*`), so released MVFNet
checkpoints load unchanged.  The forward is NOT the reference's transpose/split/Conv3d/cat pipeline: it is one
slice-only HIP stencil kernel (mvfnet_amd/csrc/mvf_*.hip) called through the C ABI, followed by `self.net`.
The Conv3d / BatchNorm3d sub-modules are parameter containers only; they are never called.
"""
import math

import torch.nn as nn

from .. import ops

__all__ = ["MVF", "make_multi_view_fusion", "HardSwish"]


class HardSwish(nn.Module):
    """x * relu6(x+3)/6 (reference: codes/models/common/se_module.py:5-24). Kept for module-tree parity; the
    MVF kernel applies it fused."""

    def forward(self, x):
        return x * nn.functional.relu6(x + 3.0) / 6.0


def _depthwise(cs, kernel):
    pad = [k // 2 for k in kernel]
    return nn.Conv3d(cs, cs, kernel, stride=1, padding=pad, groups=cs, bias=False)


class MVF(nn.Module):
    def __init__(self, net, n_segment, in_channels, alpha=0.5, use_hs=True, share=False, mode='THW'):
        super().__init__()
        if mode not in ('T', 'TH', 'THW'):
            raise ValueError("mode must be 'T', 'TH' or 'THW', got %r" % (mode,))
        self.net = net
        self.n_segment = n_segment
        self.num_shift_channel = int(in_channels * alpha)
        self.share = share
        self.use_hs = use_hs
        self.mode = mode
        cs = self.num_shift_channel
        if cs != 0:
            self.split_sizes = [cs, in_channels - cs]
            self.shift_conv = _depthwise(cs, [3, 1, 1])
            self.bn = nn.BatchNorm3d(cs)
            self.activation = HardSwish() if use_hs else nn.ReLU(inplace=True)
            if not share:
                if mode in ('TH', 'THW'):
                    self.h_conv = _depthwise(cs, [1, 3, 1])
                if mode == 'THW':
                    self.w_conv = _depthwise(cs, [1, 1, 3])
            self._initialize_weights()

    def _initialize_weights(self):
        # reference MVF.py:91-102: depthwise taps ~ N(0, sqrt(2 / (kernel_elems * out_channels))), BN (1, 0)
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                fan = m.kernel_size[0] * m.kernel_size[1] * m.kernel_size[2] * m.out_channels
                nn.init.normal_(m.weight, 0.0, math.sqrt(2.0 / fan))
            elif isinstance(m, nn.BatchNorm3d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def extra_repr(self):
        return "n_segment=%d, num_shift_channel=%d, mode=%s, share=%s, use_hs=%s" % (
            self.n_segment, self.num_shift_channel, self.mode, self.share, self.use_hs)

    def fused_input(self, x):
        """MVF-proper: the tensor the wrapped conv consumes (MVF.py:104-137)."""
        cs = self.num_shift_channel
        if cs == 0:
            return x
        if x.shape[0] % self.n_segment:
            raise ValueError("MVF: N*T = %d is not a multiple of n_segment = %d" % (x.shape[0], self.n_segment))
        bn = self.bn
        hs = self.use_hs
        training = bool(hs and (bn.training or bn.running_mean is None))
        if training and bn.running_mean is not None and bn.momentum is None:
            raise NotImplementedError("MVF: BatchNorm momentum=None (cumulative average) is not supported")
        out = ops.mvf_proper(
            x, self.shift_conv.weight, getattr(self, "h_conv", self.shift_conv).weight if not self.share else None,
            getattr(self, "w_conv", self.shift_conv).weight if not self.share else None,
            bn.weight if hs else None, bn.bias if hs else None, self.n_segment, cs, self.mode, self.share, training,
            bn.eps, bn.momentum if bn.momentum is not None else 0.1, bn.running_mean, bn.running_var)
        if training and bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1
        return out

    def forward(self, x):
        return self.net(self.fused_input(x))


def make_multi_view_fusion(net, n_segment, alpha, mvf_freq=(1, 1, 1, 1), use_hs=True, share=False, mode='THW'):
    """Wrap conv1 of every residual block of the stages selected by mvf_freq (reference MVF.py:18-49;
    n_round is 1 for every depth there, so every block is wrapped)."""
    if n_segment <= 0:
        raise ValueError("n_segment must be positive")
    for flag, name in zip(mvf_freq, ("layer1", "layer2", "layer3", "layer4")):
        if not flag:
            continue
        for block in getattr(net, name).children():
            block.conv1 = MVF(block.conv1, n_segment, block.conv1.in_channels, alpha, use_hs, share, mode)
    return net
