#!/usr/bin/env python3
"""Golden vectors for the crop / flip DECISIONS of the input pipeline, produced by the REFERENCE's own classes
(codes/datasets/pipelines/augmentations.py: ThreeCrop :465-540, CenterCrop :427-462, Flip :196-228) imported with the mmcv / cv2
placeholders of make_golden.py -- except that `mmcv.imcrop` / `mmcv.imflip` are RECORDING stand-ins here: they log the box / direction
the reference's code hands them and apply the documented mmcv 0.4.3 semantics (imcrop(img, [x1, y1, x2, y2]) = img[y1:y2+1, x1:x2+1] for an
in-bounds box; imflip(img, 'horizontal') = img[:, ::-1]) so that the classes run to completion.

What this pins (stored arrays = data only):
  * three_crop/<case>/boxes   the (3 * frames, 4) boxes in call order: which offsets, in which crop order, frame-minor
  * three_crop/<case>/order   the id (crop index * frames + frame index) of every image of the returned img_group: the oversample stacking order
  * center_crop/<case>/box    CenterCrop's box
  * flip/<seed>/...           Flip's decision for np.random.seed(seed) draws: the flag it stores in results['flip'] and whether imflip was called
What it does NOT pin: the pixel arithmetic of mmcv.imcrop / imflip themselves and of cv2 (Normalize) -- third-party code that is not in the
build container; oracle/frames_numpy.py says so.

Run in the build container: python tests/golden/make_crops_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

mg._install_stubs()
import mmcv  # noqa: E402  (the placeholder module)

LOG = []


def imcrop(img, bboxes, scale=1.0, pad_fill=None):
    b = np.asarray(bboxes).reshape(-1)
    LOG.append(("crop", tuple(int(v) for v in b)))
    x1, y1, x2, y2 = (int(v) for v in b)
    return img[y1:y2 + 1, x1:x2 + 1]


def imflip(img, direction="horizontal"):
    LOG.append(("flip", direction))
    return img[:, ::-1] if direction == "horizontal" else img[::-1]


mmcv.imcrop = imcrop
mmcv.imflip = imflip
mmcv.iminvert = lambda img: 255 - img
mmcv.is_tuple_of = lambda seq, t: isinstance(seq, tuple) and all(isinstance(v, t) for v in seq)
mmcv.is_list_of = lambda seq, t: isinstance(seq, list) and all(isinstance(v, t) for v in seq)
mmcv.impad_to_multiple = None
mmcv.imresize = None
mmcv.imrescale = None
mmcv.rescale_size = None

sys.path.insert(0, mg.REF)
from codes.datasets.pipelines.augmentations import CenterCrop, Flip, ThreeCrop  # noqa: E402

# (img_h, img_w, crop_w, crop_h): the reference's test-time recipes (256-short-side frames, 256 x 256 three-crop; 224 centre crop) and the
# two special branches (crop_h == img_h, crop_w == img_w), odd remainders included
THREE = [(256, 340, 256, 256), (256, 341, 256, 256), (340, 256, 256, 256), (343, 256, 256, 256), (300, 400, 256, 224), (301, 403, 224, 256),
         (256, 256, 256, 256), (17, 29, 8, 8)]
CENTER = [(256, 340, 224, 224), (255, 341, 224, 224), (240, 320, (200, 100))]


def marked_frames(n, h, w):
    """frame f = its index in channel 0, the row in channel 1, the column in channel 2 (uint16 -> exact)."""
    f = np.zeros((n, h, w, 3), dtype=np.int32)
    f[..., 0] = np.arange(n)[:, None, None]
    f[..., 1] = np.arange(h)[None, :, None]
    f[..., 2] = np.arange(w)[None, None, :]
    return f


out = {}
for (h, w, cw, ch) in THREE:
    tag = "three_crop/%dx%d_%dx%d" % (h, w, cw, ch)
    frames = marked_frames(3, h, w)
    del LOG[:]
    res = ThreeCrop((cw, ch))(dict(img_group=list(frames), modality="RGB"))
    boxes = np.array([e[1] for e in LOG if e[0] == "crop"], dtype=np.int64)
    out[tag + "/boxes"] = boxes
    # identify every returned image: (frame id, y0, x0) from its marks
    ident = np.array([[int(im[0, 0, 0]), int(im[0, 0, 1]), int(im[0, 0, 2]), im.shape[0], im.shape[1]] for im in res["img_group"]], dtype=np.int64)
    out[tag + "/returned"] = ident                       # rows (frame, y0, x0, h, w) in the order of results['img_group']
    out[tag + "/img_shape"] = np.array(res["img_shape"], dtype=np.int64)
for c in CENTER:
    h, w, cs = c[0], c[1], c[2] if len(c) == 3 else (c[2], c[3])
    tag = "center_crop/%dx%d_%s" % (h, w, "x".join(str(v) for v in (cs if isinstance(cs, tuple) else (cs, cs))))
    frames = marked_frames(2, h, w)
    del LOG[:]
    res = CenterCrop(cs)(dict(img_group=list(frames), modality="RGB"))
    out[tag + "/box"] = np.asarray(res["crop_bbox"], dtype=np.int64)
    out[tag + "/logged"] = np.array([e[1] for e in LOG if e[0] == "crop"], dtype=np.int64)
    out[tag + "/returned"] = np.array([[int(im[0, 0, 0]), int(im[0, 0, 1]), int(im[0, 0, 2]), im.shape[0], im.shape[1]] for im in res["img_group"]], dtype=np.int64)
for seed in (0, 1, 2, 3, 4, 5, 6, 7):
    for ratio in (0.5, 0.0, 1.0):
        tag = "flip/seed%d_ratio%g" % (seed, ratio)
        frames = marked_frames(2, 4, 6)
        np.random.seed(seed)
        del LOG[:]
        res = Flip(flip_ratio=ratio)(dict(img_group=list(frames), modality="RGB"))
        out[tag + "/flag"] = np.array(int(res["flip"]))
        out[tag + "/calls"] = np.array(sum(1 for e in LOG if e[0] == "flip"))
        out[tag + "/first_col"] = np.array(int(res["img_group"][0][0, 0, 2]))        # 0 = not mirrored, w - 1 = mirrored
np.savez_compressed(os.path.join(HERE, "crops_cases.npz"), **out)
print("wrote crops_cases.npz: %d arrays" % len(out))
