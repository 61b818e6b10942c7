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


def policy_str(base=None, **kv):
    """MVF_POLICY value: `base` (default: the current environment's) with the given switches overriding -- for child processes and for the
    library's per-call switches (csrc/common.h mvf_policy_int re-reads the variable where a test flips a switch inside one process)."""
    import os
    cur = {}
    for item in (os.environ.get("MVF_POLICY", "") if base is None else base).replace(";", ",").split(","):
        if "=" in item:
            k, v = item.split("=", 1)
            cur[k.strip().lower()] = v.strip()
    cur.update({k: str(v) for k, v in kv.items()})
    return ",".join("%s=%s" % kv_ for kv_ in cur.items())


def policy_env(**kv):
    """os.environ + MVF_POLICY with the given switches (child processes)."""
    import os
    return dict(os.environ, MVF_POLICY=policy_str(**kv))


class policy_set(object):
    """with policy_set(conv3x3_direct=0): ... -- the library's per-call switches inside this process."""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        import os
        self.old = os.environ.get("MVF_POLICY")
        os.environ["MVF_POLICY"] = policy_str(**self.kv)

    def __exit__(self, *a):
        import os
        if self.old is None:
            os.environ.pop("MVF_POLICY", None)
        else:
            os.environ["MVF_POLICY"] = self.old


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
    """Context manager: oracle/net_torch.py with the engine's bf16 STORAGE emulated on the CPU -- every tensor the HIP engine keeps
    in HBM as bf16 (stem input, packed conv weights, conv outputs z, ReLU / block / MVF outputs) is rounded to bf16 at the
    oracle's storage hooks (net_torch.Q); arithmetic, BatchNorm statistics, the head and all parameter gradients stay fp32.
    With backward=True the GRADIENTS stored as bf16 are rounded at the same points during autograd's backward (dz of every
    BatchNorm backward, the data gradients da / dx, the block-output gradient g after the skip-connection add, the stem
    pooling gradient): the emulation then rounds wherever the engine rounds, so engine and emulation differ only by fp32
    summation order -- a few bf16 ulps flipping -- instead of the O(10 %) bf16-vs-fp32 gap of the synthetic network."""

    def __init__(self, backward=False):
        self.backward = backward

    def __enter__(self):
        import torch
        from oracle import net_torch

        def r(t):
            return t.bfloat16().float()

        rb = self.backward

        class _Both(torch.autograd.Function):         # stored tensor whose gradient is stored too
            @staticmethod
            def forward(ctx, t):
                return r(t)

            @staticmethod
            def backward(ctx, g):
                return r(g) if rb else g

        class _Fwd(torch.autograd.Function):          # rounded operand, fp32 gradient (weights)
            @staticmethod
            def forward(ctx, t):
                return r(t)

            @staticmethod
            def backward(ctx, g):
                return g

        class _Bwd(torch.autograd.Function):          # only the gradient is stored
            @staticmethod
            def forward(ctx, t):
                return t.view_as(t)

            @staticmethod
            def backward(ctx, g):
                return r(g) if rb else g

        class _Q(object):
            x = staticmethod(lambda t: _Fwd.apply(t))
            w = staticmethod(lambda t: _Fwd.apply(t))
            z = staticmethod(lambda t: _Both.apply(t))
            a = staticmethod(lambda t: _Both.apply(t))
            gb = staticmethod(lambda t: _Bwd.apply(t))
            y = staticmethod(lambda t: _Both.apply(t))
            i = staticmethod(lambda t: t)

        self._mod, self._old = net_torch, net_torch.Q
        net_torch.Q = _Q()
        return self

    def __exit__(self, *a):
        self._mod.Q = self._old


class bf16_inference_oracle(object):
    """oracle/net_torch.py as the bf16 INFERENCE engine stores things: BatchNorm folded into the conv weights BEFORE they are
    rounded to bf16 (use fold_bn_state_dict on the state_dict first), conv epilogue = fp32 accumulate + shift (+ residual) +
    ReLU, ONE rounding of the stored activation; the MVF slice is rounded once after BN + hard-swish."""

    def __enter__(self):
        from oracle import net_torch

        def r(t):
            return t.bfloat16().float()

        class _Q(object):
            x = w = a = i = staticmethod(r)
            z = gb = y = staticmethod(lambda t: t)

        self._mod, self._old = net_torch, net_torch.Q
        net_torch.Q = _Q()
        return self

    def __exit__(self, *a):
        self._mod.Q = self._old


def fold_bn_state_dict(sd, eps=1e-5):
    """Eval-mode state_dict with every conv's BatchNorm scale folded into the conv weight (fp32, the formula of mvf_bn_fold:
    s = gamma / sqrt(var + eps), shift = beta - mean * s) and the BatchNorm replaced by `+ shift` (gamma = 1, mean = 0,
    var = 1 - eps): same function, but the weights a bf16 engine rounds are the folded ones.  MVF BatchNorms are left alone."""
    import torch
    out = {k: v.detach().clone() for k, v in sd.items()}
    pairs = [("backbone.conv1.weight", "backbone.bn1.")]
    for k in sd:
        if k.endswith("conv2.weight"):
            pre = k[: -len("conv2.weight")]
            c1 = pre + ("conv1.net.weight" if (pre + "conv1.net.weight") in sd else "conv1.weight")
            pairs += [(c1, pre + "bn1."), (pre + "conv2.weight", pre + "bn2."), (pre + "conv3.weight", pre + "bn3.")]
            if (pre + "downsample.0.weight") in sd:
                pairs.append((pre + "downsample.0.weight", pre + "downsample.1."))
    for wk, bn in pairs:
        g, b, m, v = (sd[bn + n].detach().float() for n in ("weight", "bias", "running_mean", "running_var"))
        s = g / torch.sqrt(v + eps)
        out[wk] = sd[wk].detach() * s.view(-1, 1, 1, 1)
        out[bn + "weight"] = torch.ones_like(g)
        out[bn + "bias"] = b - m * s
        out[bn + "running_mean"] = torch.zeros_like(m)
        out[bn + "running_var"] = torch.full_like(v, 1.0 - eps)
    return out
