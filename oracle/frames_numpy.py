"""CPU restatement of the input pipeline's device-side part: crop window -> horizontal flip -> Normalize -> FormatShape('NCHW').

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg) -- never imported by mvfnet_amd.

Follows, step by step:
  * crop      codes/datasets/pipelines/augmentations.py:465-540 (ThreeCrop offsets; `mmcv.imcrop(img, [x0, y0, x0+w-1, y0+h-1])`
              = img[y0:y0+h, x0:x0+w] for an in-bounds box) and CenterCrop
  * flip      augmentations.py:196-228 (`mmcv.imflip(img, 'horizontal')` = img[:, ::-1])
  * normalize augmentations.py:365-374: float32(img); BGR->RGB when to_rgb; subtract mean, then multiply by 1/float64(std),
              each step rounded to float32 (cv2.subtract / cv2.multiply on a CV_32F image with a scalar)
  * stacking  codes/datasets/pipelines/formating.py:146-160: per frame HWC -> CHW, np.stack over frames

PARITY STATUS: FormatShape is pinned by a golden vector from the reference's own class (tests/golden/make_frames_golden.py).
[r3] The crop / flip DECISIONS -- ThreeCrop's three boxes and their order, the crop-major / frame-minor order of the oversampled
group, CenterCrop's box, Flip's `np.random.rand() < flip_ratio` draw -- are the reference's own code and are pinned by
tests/golden/crops_cases.npz, recorded from those classes through logging stand-ins for `mmcv.imcrop` / `mmcv.imflip`
(tests/golden/make_crops_golden.py; tests/test_frames_cpu.py).  What stays **parity unpinned** is third-party pixel arithmetic only:
`mmcv.imcrop` / `imflip` themselves (mmcv 0.4.3: a slice, a reversed view) and cv2's cvtColor / subtract / multiply inside Normalize
(restated from their documented fp32 behaviour) -- neither library is installed in the build container."""
import numpy as np


def three_crop_offsets(img_h, img_w, crop_h, crop_w):
    """(x0, y0) of the three crops in the reference's order (augmentations.py:487-510)."""
    if crop_h == img_h:
        s = (img_w - crop_w) // 2
        return [(0, 0), (2 * s, 0), (s, 0)]
    if crop_w == img_w:
        s = (img_h - crop_h) // 2
        return [(0, 0), (0, 2 * s), (0, s)]
    ws, hs = (img_w - crop_w) // 4, (img_h - crop_h) // 4
    return [(0, 2 * hs), (4 * ws, 2 * hs), (2 * ws, 2 * hs)]


def three_crop_windows(n_frames, img_h, img_w, crop_h, crop_w):
    """(3 * n_frames, 3) rows (y0, x0, flip = 0) in the order of ThreeCrop's returned img_group (augmentations.py:512-530: for every
    offset all frames, the flipped copies are built and dropped)."""
    return np.array([(y0, x0, 0) for (x0, y0) in three_crop_offsets(img_h, img_w, crop_h, crop_w) for _ in range(n_frames)], dtype=np.int32)


def center_crop_offset(img_h, img_w, crop_h, crop_w):
    """(x0, y0) of CenterCrop (augmentations.py:447-452)."""
    return (img_w - crop_w) // 2, (img_h - crop_h) // 2


def flip_decision(draw, flip_ratio):
    """Flip.__call__ (augmentations.py:217): one uniform draw per sample, mirrored when it is below the ratio."""
    return bool(draw < flip_ratio)


def imnormalize(img_u8_hwc, mean, std, to_rgb, div_255=False):
    img = np.float32(img_u8_hwc)
    if div_255:
        img = img / np.float32(255)
    if to_rgb:
        img = img[..., ::-1]
    mean32 = np.float32(np.float64(np.asarray(mean, dtype=np.float32)))
    stdinv32 = np.float32(1.0 / np.float64(np.asarray(std, dtype=np.float32)))
    out = (img - mean32).astype(np.float32)
    return (out * stdinv32).astype(np.float32)


def frames_to_nchw(frames_u8, window, h, w, mean, std, to_rgb=True, div_255=False):
    """frames_u8 (n, hs, ws, 3) uint8, window (n,3) rows (y0, x0, flip) or None -> (n,3,h,w) float32."""
    n = frames_u8.shape[0]
    out = np.empty((n, 3, h, w), dtype=np.float32)
    for i in range(n):
        y0, x0, flip = (0, 0, 0) if window is None else (int(window[i, 0]), int(window[i, 1]), int(window[i, 2]))
        img = frames_u8[i, y0:y0 + h, x0:x0 + w]
        if flip:
            img = img[:, ::-1]
        out[i] = imnormalize(img, mean, std, to_rgb, div_255).transpose(2, 0, 1)
    return out
