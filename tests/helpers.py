"""Shared helpers for the parity tests."""
import os

import numpy as np

from mvfnet_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_cache = {}


def golden(name):
    if name not in _cache:
        _cache[name] = np.load(os.path.join(GOLDEN, name))
    return _cache[name]


def rel_err(a, ref):
    """max|a-ref| / max|ref| -- tolerances are relative to each tensor's scale."""
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert a.shape == ref.shape, (a.shape, ref.shape)
    d = float(np.abs(a - ref).max()) if a.size else 0.0
    s = float(np.abs(ref).max()) if ref.size else 0.0
    return d / max(s, 1e-30) if s > 0 else d


def mvf_case_params(name, C, alpha, mode, share, use_hs, planes, net_kind):
    """The synth state_dict of one MVF golden case, keyed like the reference module's state_dict."""
    cs = int(C * alpha)
    shapes = {}
    if net_kind == "conv":
        shapes["net.weight"] = (planes, C, 1, 1)
    if cs:
        shapes["shift_conv.weight"] = (cs, 1, 3, 1, 1)
        for k in ("weight", "bias", "running_mean", "running_var"):
            shapes["bn." + k] = (cs,)
        shapes["bn.num_batches_tracked"] = ()
        if not share:
            if mode in ("TH", "THW"):
                shapes["h_conv.weight"] = (cs, 1, 1, 3, 1)
            if mode == "THW":
                shapes["w_conv.weight"] = (cs, 1, 1, 1, 3)
    pre = "mvf/%s/" % name
    vals = synth.synth_state_dict({pre + k: v for k, v in shapes.items()})
    return {k: vals[pre + k] for k in shapes}


def rel_l2(a, ref):
    """||a-ref||_2 / ||ref||_2 -- for gradients under bf16 storage: a ReLU mask that flips on a near-zero activation moves
    single elements by O(1) (max-norm useless) while leaving the gradient's energy essentially unchanged."""
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert a.shape == ref.shape, (a.shape, ref.shape)
    return float(np.sqrt(((a - ref) ** 2).sum()) / max(np.sqrt((ref ** 2).sum()), 1e-30))


class bf16_storage_oracle(object):
    """Context manager: oracle/net_torch.py with bf16 STORAGE emulated -- conv inputs / weights / outputs and ReLU outputs are
    rounded to bf16, arithmetic stays fp32 (what the engine's bf16 mode keeps in HBM).  On the synthetic network bf16 storage
    alone moves layer4 by ~40 % in relative L2 (random weights, 16 residual blocks); against THIS yardstick the engine must
    agree to the rounding-point level."""

    def __enter__(self):
        import torch.nn.functional as F
        from oracle import net_torch
        real = F

        def r(t):
            return t.bfloat16().float()

        class _F(object):
            def __getattr__(self, name):
                return getattr(real, name)

            @staticmethod
            def conv2d(x, w, *a, **k):
                return r(real.conv2d(r(x), r(w), *a, **k))

            @staticmethod
            def relu(x):
                return r(real.relu(x))

        self._mod, self._old = net_torch, net_torch.F
        net_torch.F = _F()
        return self

    def __exit__(self, *a):
        self._mod.F = self._old
