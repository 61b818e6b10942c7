"""ORACLE (test infrastructure, not product code): CPU restatement of the MVF module.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
The product path (mvfnet_amd/) never does; it fails loudly without the HIP library.

What is restated (reference = /root/reference, whwu95/MVFNet):
  * MVF.forward                      codes/models/modules/MVF.py:104-138
  * HardSwish = x * relu6(x+3)/6     codes/models/common/se_module.py:5-24
  * BatchNorm3d(Cs) semantics        codes/models/modules/MVF.py:69,133 (torch defaults:
                                     eps 1e-5, momentum 0.1, biased var to normalise,
                                     unbiased var into running_var)
The reference formulation (view -> transpose -> split -> three depthwise Conv3d -> add -> BN3d ->
hswish -> cat -> transpose -> contiguous) is restated in closed form on the (N*T, C, H, W) tensor;
no transposes are needed because frame t+-1 of the same clip is the neighbouring image.

Pinned: tests/test_oracle_golden.py checks every function here against tests/golden/mvf_cases.npz,
which tests/golden/make_golden.py produced by running the imported reference module (fwd, all
grads through torch autograd, running stats) on formula-generated inputs.

The arithmetic of Conv3d/BatchNorm3d lives in PyTorch ATen (not under /root/reference; the
reference pins "PyTorch 1.5" in README.md:28 only; this container has torch 2.10). The reference
has no tests of its own for this path -- the golden vectors above are the pin.
"""
import numpy as np

EPS = 1e-5
MOMENTUM = 0.1


def _shift(a, axis, d):
    """b[..., i, ...] = a[..., i + d, ...] with zero fill (the Conv3d zero padding, MVF.py:66,77,80)."""
    b = np.zeros_like(a)
    n = a.shape[axis]
    if abs(d) >= n:
        return b
    src = [slice(None)] * a.ndim
    dst = [slice(None)] * a.ndim
    if d > 0:
        src[axis] = slice(d, n)
        dst[axis] = slice(0, n - d)
    elif d < 0:
        src[axis] = slice(0, n + d)
        dst[axis] = slice(-d, n)
    b[tuple(dst)] = a[tuple(src)]
    return b


def _views(mode, share, wt, wh, ww):
    """List of (axis in (N,T,Cs,H,W), weight (Cs,3), name-of-grad-slot). MVF.py:112-129."""
    v = [(1, wt, "t")]
    if mode in ("TH", "THW"):
        v.append((3, wt if share else wh, "t" if share else "h"))
    if mode == "THW":
        v.append((4, wt if share else ww, "t" if share else "w"))
    if mode not in ("T", "TH", "THW"):
        raise ValueError("mode must be 'T', 'TH' or 'THW'")
    return v


def hswish(u):
    return u * np.clip(u + 3.0, 0.0, 6.0) / 6.0


def hswish_grad(u):
    """d/du [u * relu6(u+3)/6]; relu6' is 0 AT the kinks (torch hardtanh_backward: strict inequalities)."""
    inner = ((u > -3.0) & (u < 3.0)).astype(u.dtype)
    return np.clip(u + 3.0, 0.0, 6.0) / 6.0 + u * inner / 6.0


def mvf_forward(x, n_segment, cs, wt, wh=None, ww=None, mode="THW", share=False, use_hs=True,
                gamma=None, beta=None, running_mean=None, running_var=None, training=False,
                dtype=np.float64):
    """MVF-proper (everything in MVF.forward except self.net), MVF.py:104-137.

    x: (NT, C, H, W). Returns (out (NT,C,H,W), cache, (new_running_mean, new_running_var)).
    Weights are (Cs, 3): tap j multiplies the element at offset j-1 along the view axis
    (cross-correlation, as torch Conv3d).
    """
    x = np.asarray(x, dtype=dtype)
    nt, c, h, w = x.shape
    if nt % n_segment:
        raise ValueError("NT must be a multiple of n_segment (MVF.py:107-109 view)")
    if cs == 0:
        return x.copy(), None, (running_mean, running_var)
    n = nt // n_segment
    s = x.reshape(n, n_segment, c, h, w)[:, :, :cs]                       # (N,T,Cs,H,W) slice
    y = np.zeros_like(s)
    for axis, wgt, _ in _views(mode, share, wt, wh, ww):
        wgt = np.asarray(wgt, dtype=dtype).reshape(cs, 3)
        for j in range(3):
            y += wgt[None, None, :, j, None, None] * _shift(s, axis, j - 1)
    cache = dict(s=s, n_segment=n_segment, cs=cs, mode=mode, share=share, use_hs=use_hs, training=training,
                 wt=wt, wh=wh, ww=ww, shape=x.shape)
    new_rm, new_rv = running_mean, running_var
    if use_hs:
        gamma = np.asarray(gamma, dtype=dtype)
        beta = np.asarray(beta, dtype=dtype)
        if training:
            m = n * n_segment * h * w
            mean = y.mean(axis=(0, 1, 3, 4))
            var = y.var(axis=(0, 1, 3, 4))                                   # biased
            new_rm = (1 - MOMENTUM) * np.asarray(running_mean, dtype) + MOMENTUM * mean
            new_rv = (1 - MOMENTUM) * np.asarray(running_var, dtype) + MOMENTUM * var * m / max(m - 1, 1)
        else:
            mean = np.asarray(running_mean, dtype=dtype)
            var = np.asarray(running_var, dtype=dtype)
        invstd = 1.0 / np.sqrt(var + EPS)
        xhat = (y - mean[None, None, :, None, None]) * invstd[None, None, :, None, None]
        u = xhat * gamma[None, None, :, None, None] + beta[None, None, :, None, None]
        o = hswish(u)
        cache.update(xhat=xhat, u=u, invstd=invstd, gamma=gamma)
    else:
        o = y
    out = x.reshape(n, n_segment, c, h, w).copy()
    out[:, :, :cs] = o
    return out.reshape(nt, c, h, w), cache, (new_rm, new_rv)


def mvf_backward(g, cache):
    """Backward of mvf_forward (the reference relies on autograd).

    g: (NT,C,H,W) grad w.r.t. mvf_forward's output. Returns dict(dx, dwt, dwh, dww, dgamma, dbeta).
    """
    dtype = cache["s"].dtype
    g = np.asarray(g, dtype=dtype)
    nt, c, h, w = cache["shape"]
    T, cs = cache["n_segment"], cache["cs"]
    n = nt // T
    g5 = g.reshape(n, T, c, h, w)
    go = g5[:, :, :cs]
    res = dict(dgamma=None, dbeta=None)
    if cache["use_hs"]:
        du = go * hswish_grad(cache["u"])
        gamma, invstd, xhat = cache["gamma"], cache["invstd"], cache["xhat"]
        res["dbeta"] = du.sum(axis=(0, 1, 3, 4))
        res["dgamma"] = (du * xhat).sum(axis=(0, 1, 3, 4))
        gi = (gamma * invstd)[None, None, :, None, None]
        if cache["training"]:
            m = n * T * h * w
            dy = gi * (du - res["dbeta"][None, None, :, None, None] / m
                       - xhat * res["dgamma"][None, None, :, None, None] / m)
        else:
            dy = gi * du
    else:
        dy = go
    s = cache["s"]
    ds = np.zeros_like(s)
    dw = dict(t=np.zeros((cs, 3), dtype), h=np.zeros((cs, 3), dtype), w=np.zeros((cs, 3), dtype))
    for axis, wgt, slot in _views(cache["mode"], cache["share"], cache["wt"], cache["wh"], cache["ww"]):
        wgt = np.asarray(wgt, dtype=dtype).reshape(cs, 3)
        for j in range(3):
            dw[slot][:, j] += (dy * _shift(s, axis, j - 1)).sum(axis=(0, 1, 3, 4))
            ds += wgt[None, None, :, j, None, None] * _shift(dy, axis, -(j - 1))
    dx = g5.copy()
    dx[:, :, :cs] = ds
    res.update(dx=dx.reshape(nt, c, h, w), dwt=dw["t"], dwh=dw["h"], dww=dw["w"])
    return res
