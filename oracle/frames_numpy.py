"""CPU restatement of the input pipeline's device-side part: crop window -> horizontal flip -> Normalize -> FormatShape('NCHW').

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg) -- never imported by mvfnet_amd.

Follows, step by step:
  * crop      codes/datasets/pipelines/augmentations.py:465-540 (ThreeCrop offsets; `mmcv.imcrop(img, [x0, y0, x0+w-1, y0+h-1])`
              = img[y0:y0+h, x0:x0+w] for an in-bounds box) and CenterCrop
  * flip      augmentations.py:196-228 (`mmcv.imflip(img, 'horizontal')` = img[:, ::-1])
  * normalize augmentations.py:365-374: float32(img); BGR->RGB when to_rgb; subtract mean, then multiply by 1/float64(std),
              each step rounded to float32 (cv2.subtract / cv2.multiply on a CV_32F image with a scalar)
  * stacking  codes/datasets/pipelines/formating.py:146-160: per frame HWC -> CHW, np.stack over frames

PARITY STATUS: every step that is the REFERENCE's own code is pinned by golden vectors recorded from its own classes:
  * FormatShape                      tests/golden/make_frames_golden.py (frames_cases.npz)
  * ThreeCrop / CenterCrop / Flip    [r3] boxes, crop order, oversample stacking order, the flip draw -- recorded through logging stand-ins
                                     for `mmcv.imcrop` / `mmcv.imflip` (make_crops_golden.py, crops_cases.npz)
  * SampleFrames                     [r4] `sample_frame_inds` below against the reference's SampleFrames for 69 (case, seed) pairs: the three
                                     branches of _sample_clips, _test_sample_clips for sth_samples 1 / 2 / 10 / generic, temporal jitter,
                                     the `minimum(total_frames - 1)` clamp, the shipped 8x8 / 16x4 / 10-clip test recipes, and the NUMBER
                                     of random draws (make_sampling_golden.py, sampling_cases.npz)
  * Normalize                        [r4] `imnormalize` below against the reference's Normalize (augmentations.py:343-386 is the reference's own
                                     class): call order cvtColor -> subtract -> multiply, in place, the float64 (1,3) scalar operands
                                     float64(float32(mean)) and 1 / float64(float32(std)), `div_255` (uint8 / 255 in fp32 before the call),
                                     `to_rgb`, img_norm_cfg -- recorded through logging stand-ins for the three cv2 primitives
                                     (make_normalize_golden.py, normalize_cases.npz); outputs bit-equal.
What stays **parity unpinned** is third-party pixel arithmetic only: `mmcv.imcrop` / `imflip` themselves (mmcv 0.4.3: a slice, a reversed
view) and cv2's cvtColor / subtract / multiply on a CV_32F image with a scalar operand (one rounded fp32 operation per element, the scalar
converted to fp32) -- restated from their documented behaviour; neither library is installed in the build container."""
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


def sample_frame_inds(total_frames, clip_len, frame_interval, num_clips, test_mode, temporal_jitter=False, sth_samples=1, rng=np.random):
    """SampleFrames._get_frame_inds (codes/datasets/pipelines/loading.py:96-116) with _sample_clips (:35-60) and _test_sample_clips
    (:62-94) written out statement by statement; `rng` supplies `randint` (the reference uses the global np.random)."""
    ori_clip_len = clip_len * frame_interval

    def sample_clips():                                            # loading.py:47-60
        avg_interval = (total_frames - ori_clip_len + 1) // num_clips
        if avg_interval > 0:
            return np.arange(num_clips) * avg_interval + rng.randint(avg_interval, size=num_clips)
        if total_frames > max(num_clips, ori_clip_len):
            return np.sort(rng.randint(total_frames - ori_clip_len + 1, size=num_clips))
        return np.zeros((num_clips,))

    if test_mode:                                                  # loading.py:62-94
        tick = (total_frames - ori_clip_len + 1) / float(num_clips)
        if sth_samples == 1:
            clip_offsets = np.array([int(tick / 2.0 + tick * x) for x in range(num_clips)]) if tick > 0 else np.zeros((num_clips,))
        elif sth_samples == 2:
            clip_offsets = np.array([int(tick / 2.0 + tick * x) for x in range(num_clips)] + [int(tick * x) for x in range(num_clips)])
        elif sth_samples == 10:
            offsets = []
            for _ in range(10):
                offsets += sample_clips().tolist()
            clip_offsets = np.array(offsets)
        else:
            rows = [np.array([int(tick / 2.0 + tick * x) for x in range(num_clips)])]
            avg_duration = (total_frames - ori_clip_len + 1) // float(num_clips)
            for _ in range(sth_samples - 1):
                rows.append(np.multiply(list(range(num_clips)), avg_duration) + rng.randint(avg_duration, size=num_clips))
            clip_offsets = np.stack(rows).reshape(-1)
    else:
        clip_offsets = sample_clips()
    frame_inds = clip_offsets[:, None] + np.arange(clip_len)[None, :] * frame_interval          # loading.py:103-104
    if temporal_jitter:                                                                          # :105-110 one draw, shared by the clips
        frame_inds = frame_inds + rng.randint(frame_interval, size=clip_len)[None, :]
    frame_inds = np.concatenate(frame_inds)
    return np.minimum(frame_inds, total_frames - 1).astype(np.int64)                             # :115
