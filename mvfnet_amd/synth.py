"""Deterministic, formula-generated weights and inputs.

An MVFNet-R50 state_dict is ~97 MB, so fixtures never store weights or inputs.
Instead every tensor is a pure function of (key name, shape, seed): the golden
generator (tests/golden/make_golden.py, run once next to the imported reference)
and every parity test on the GPU box call the SAME functions below, so both
sides see bit-identical fp32 values without shipping them.

numpy's legacy ``RandomState`` stream (MT19937 + frozen ``standard_normal`` /
``random_sample``) is stable across numpy versions and platforms.
"""
import zlib

import numpy as np

__all__ = ["rng_for", "synth_tensor", "synth_state_dict", "synth_clip_batch", "synth_labels"]


def rng_for(key, seed=0):
    """RandomState keyed by a string: crc32(key) xor a seed."""
    return np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)


def synth_tensor(key, shape, kind="normal", scale=1.0, shift=0.0, seed=0):
    """fp32 ndarray of `shape`; kind 'normal' -> N(shift, scale), 'uniform' -> U[shift, shift+scale)."""
    r = rng_for(key, seed)
    n = int(np.prod(shape)) if len(shape) else 1
    if kind == "normal":
        a = r.standard_normal(n) * scale + shift
    elif kind == "uniform":
        a = r.random_sample(n) * scale + shift
    else:
        raise ValueError(kind)
    return a.astype(np.float32).reshape(shape)


def _fan_in(shape):
    f = 1
    for s in shape[1:]:
        f *= s
    return max(f, 1)


def synth_state_dict(shapes, seed=0, fc_std=0.05):
    """Formula-generated values for every entry of a model state_dict.

    `shapes` maps state_dict key -> shape tuple (from ``model.state_dict()``).
    Rules (chosen so activations stay O(1) through 50/101 layers and BN folding
    is actually exercised -- SURVEY.md Appendix E):
      * conv / linear weights (ndim >= 2): N(0, sqrt(2/fan_in)) (He), fc: N(0, fc_std)
      * BN weight (gamma): U[0.5, 1.0); the LAST BN of a bottleneck (bn3) and the
        downsample BN get U[0.25, 0.5) so the residual sum does not blow up
      * BN bias (beta): N(0, 0.1)
      * running_mean: N(0, 0.1); running_var: U[0.5, 1.5)
      * num_batches_tracked: 0 (int64)
      * linear bias: N(0, 0.01)
    """
    out = {}
    for key, shape in shapes.items():
        shape = tuple(shape)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[key] = np.zeros(shape, dtype=np.int64)
        elif leaf == "running_mean":
            out[key] = synth_tensor(key, shape, "normal", 0.1, 0.0, seed)
        elif leaf == "running_var":
            out[key] = synth_tensor(key, shape, "uniform", 1.0, 0.5, seed)
        elif leaf == "weight" and len(shape) == 1:
            last = (".bn3." in key) or (".downsample.1." in key)
            lo, width = (0.25, 0.25) if last else (0.5, 0.5)
            out[key] = synth_tensor(key, shape, "uniform", width, lo, seed)
        elif leaf == "bias" and ("bn" in key or "downsample.1" in key):
            out[key] = synth_tensor(key, shape, "normal", 0.1, 0.0, seed)
        elif leaf == "bias":
            out[key] = synth_tensor(key, shape, "normal", 0.01, 0.0, seed)
        elif leaf == "weight" and len(shape) == 2:
            out[key] = synth_tensor(key, shape, "normal", fc_std, 0.0, seed)
        elif leaf == "weight":
            out[key] = synth_tensor(key, shape, "normal", float(np.sqrt(2.0 / _fan_in(shape))), 0.0, seed)
        else:
            raise KeyError("synth_state_dict: no rule for %s %s" % (key, shape))
    return out


def synth_clip_batch(n_clips, t, h, w, seed=0, c=3):
    """[B, T, 3, H, W] fp32 ~ N(0,1): the post-Normalize range of the reference pipeline."""
    return synth_tensor("img_group", (n_clips, t, c, h, w), "normal", 1.0, 0.0, seed)


def synth_labels(n_clips, num_classes=400, seed=0):
    r = rng_for("labels", seed)
    return r.randint(0, num_classes, size=(n_clips, 1)).astype(np.int64)
