#!/usr/bin/env python3
"""Golden vectors for Normalize, produced by the REFERENCE's own class (codes/datasets/pipelines/augmentations.py:343-386 -- the
reference's code, not mmcv's) imported with the placeholders of make_golden.py.  Only the three cv2 primitives it calls are third-party
and absent from the build container; they are RECORDING numpy stand-ins here that log every call (function, order, the scalar operand's
dtype / shape / values, whether the destination is the source = in place) and apply OpenCV's documented CV_32F semantics so that the class
runs to completion:
  cv2.cvtColor(img, COLOR_BGR2RGB, img)  -> channel order reversed in place
  cv2.subtract(img, scalar, img)         -> img - float32(scalar) per channel, one rounded fp32 operation (arithm_op converts a
                                            scalar operand to the working depth of a CV_32F source)
  cv2.multiply(img, scalar, img)         -> img * float32(scalar), likewise

What this pins (stored arrays = data only):
  <case>/calls      the call sequence as codes (0 cvtColor, 1 subtract, 2 multiply) per image: cvtColor BEFORE subtract BEFORE multiply
  <case>/sub_scalar the operand handed to cv2.subtract: float64, shape (1, 3), = float64(float32(mean))
  <case>/mul_scalar the operand handed to cv2.multiply: float64, shape (1, 3), = 1 / float64(float32(std))
  <case>/inplace    every call wrote into its own source
  <case>/in_dtype   dtype code of the image entering imnormalize (div_255: already float32 = uint8 / 255 in fp32)
  <case>/out        the normalised images (fp32, HWC) the class returned; <case>/frames the uint8 input
  <case>/cfg        img_norm_cfg as written to results
What it does NOT pin: cv2's own arithmetic (restated above from its documentation; oracle/frames_numpy.py says so).

Run in the build container: python tests/golden/make_normalize_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

mg._install_stubs()
import cv2  # noqa: E402  (the placeholder module)
import mmcv  # noqa: E402

LOG = []


def _scalar(s):
    s = np.asarray(s)
    return s, np.float32(s.reshape(-1))          # OpenCV converts the scalar to the CV_32F working type


def cvtColor(src, code, dst=None):
    LOG.append(("cvtColor", int(code), dst is src, None))
    out = src[..., ::-1].copy()
    if dst is not None:
        dst[...] = out
        return dst
    return out


def subtract(src1, src2, dst=None):
    raw, s32 = _scalar(src2)
    LOG.append(("subtract", raw.copy(), dst is src1, str(raw.dtype)))
    out = (src1 - s32).astype(np.float32)
    if dst is not None:
        dst[...] = out
        return dst
    return out


def multiply(src1, src2, dst=None):
    raw, s32 = _scalar(src2)
    LOG.append(("multiply", raw.copy(), dst is src1, str(raw.dtype)))
    out = (src1 * s32).astype(np.float32)
    if dst is not None:
        dst[...] = out
        return dst
    return out


cv2.COLOR_BGR2RGB = 4
cv2.cvtColor, cv2.subtract, cv2.multiply = cvtColor, subtract, multiply
for name in ("imcrop", "imflip", "iminvert", "impad_to_multiple", "imresize", "imrescale", "rescale_size"):
    setattr(mmcv, name, None)
mmcv.is_tuple_of = lambda seq, t: isinstance(seq, tuple) and all(isinstance(v, t) for v in seq)
mmcv.is_list_of = lambda seq, t: isinstance(seq, list) and all(isinstance(v, t) for v in seq)

sys.path.insert(0, mg.REF)
from codes.datasets.pipelines.augmentations import Normalize  # noqa: E402

K400 = ([123.675, 116.28, 103.53], [58.395, 57.12, 57.375])          # configs/MVFNet/K400/*.py img_norm_cfg, to_rgb=True
UNIT = ([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])                # the div_255 convention
# (tag, mean, std, div_255, to_rgb, (n, h, w), seed)
CASES = [
    ("k400_rgb", K400[0], K400[1], False, True, (3, 9, 11), 1),
    ("k400_bgr", K400[0], K400[1], False, False, (2, 8, 8), 2),
    ("div255_rgb", UNIT[0], UNIT[1], True, True, (2, 7, 5), 3),
    ("div255_bgr", UNIT[0], UNIT[1], True, False, (1, 4, 6), 4),
    ("all_values", K400[0], K400[1], False, True, (1, 16, 16), None),  # every uint8 value in every channel
]
CODE = {"cvtColor": 0, "subtract": 1, "multiply": 2}
DT = {"uint8": 0, "float32": 1, "float64": 2}

out = {}
for (tag, mean, std, div, rgb, (n, h, w), seed) in CASES:
    if seed is None:
        fr = np.arange(256, dtype=np.uint8).reshape(1, 16, 16, 1).repeat(3, axis=3).copy()
        fr[..., 1] = fr[..., 1][:, ::-1]
        fr[..., 2] = fr[..., 2][:, :, ::-1]
    else:
        fr = np.random.RandomState(seed).randint(0, 256, size=(n, h, w, 3)).astype(np.uint8)
    del LOG[:]
    norm = Normalize(mean, std, div_255=div, to_rgb=rgb)
    in_dtypes = []
    orig = norm.imnormalize

    def spy(img, mean_, std_, to_rgb=True, _orig=orig):
        in_dtypes.append(DT[str(img.dtype)])
        return _orig(img, mean_, std_, to_rgb)

    norm.imnormalize = spy
    res = norm(dict(img_group=[f for f in fr]))
    imgs = np.stack(res["img_group"])
    assert imgs.dtype == np.float32
    out[tag + "/frames"] = fr
    out[tag + "/out"] = imgs
    out[tag + "/calls"] = np.array([CODE[e[0]] for e in LOG], dtype=np.int64)
    subs = [e for e in LOG if e[0] == "subtract"]
    muls = [e for e in LOG if e[0] == "multiply"]
    assert all(e[3] == "float64" and e[1].shape == (1, 3) for e in subs + muls)
    out[tag + "/sub_scalar"] = subs[0][1]
    out[tag + "/mul_scalar"] = muls[0][1]
    out[tag + "/scalars_constant"] = np.array(int(all(np.array_equal(e[1], subs[0][1]) for e in subs) and all(np.array_equal(e[1], muls[0][1]) for e in muls)))
    out[tag + "/inplace"] = np.array(int(all(e[2] for e in LOG)))
    out[tag + "/in_dtype"] = np.array(in_dtypes, dtype=np.int64)
    cfg = res["img_norm_cfg"]
    assert cfg["mean"].dtype == np.float32 and cfg["std"].dtype == np.float32
    out[tag + "/cfg_mean"], out[tag + "/cfg_std"] = cfg["mean"], cfg["std"]
    out[tag + "/cfg_flags"] = np.array([int(cfg["div_255"]), int(cfg["to_rgb"])], dtype=np.int64)
np.savez_compressed(os.path.join(HERE, "normalize_cases.npz"), **out)
print("wrote normalize_cases.npz: %d arrays" % len(out))
