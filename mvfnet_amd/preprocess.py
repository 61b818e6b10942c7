"""Device side of the input pipeline: the host ships DECODED uint8 frames (a quarter of the bytes of the
normalised fp32 tensor the reference's DataLoader + scatter move, codes/core/parallel/distributed.py:40-62) and one HIP kernel
does crop window -> flip -> Normalize -> FormatShape -> stem layout (`mvf_frames_prep_u8`, include/mvfnet_hip.h).

Mirrors the reference's pipeline steps in names and argument meaning: `img_norm_cfg = dict(mean, std, to_rgb)` of the configs
(config_zoo R50 8x8: mean [123.675, 116.28, 103.53], std [58.395, 57.12, 57.375], to_rgb True), `Flip(flip_ratio)`'s boolean,
`CenterCrop` / `ThreeCrop(crop_size)` offsets (augmentations.py:196-228, 342-396, 465-540).  Decoding, resizing and the random
scale-jitter crop stay on the host."""
import ctypes

import torch

from ._lib import check, lib

_DT = {torch.float32: 0, torch.bfloat16: 1}


def three_crop_offsets(img_h, img_w, crop_h, crop_w):
    """(x0, y0) of ThreeCrop's crops in the reference's order: left/top, right/bottom, middle (augmentations.py:487-510)."""
    if crop_h == img_h:
        s = (img_w - crop_w) // 2
        return [(0, 0), (2 * s, 0), (s, 0)]
    if crop_w == img_w:
        s = (img_h - crop_h) // 2
        return [(0, 0), (0, 2 * s), (0, s)]
    ws, hs = (img_w - crop_w) // 4, (img_h - crop_h) // 4
    return [(0, 2 * hs), (4 * ws, 2 * hs), (2 * ws, 2 * hs)]


def three_crop_windows(n_frames, img_h, img_w, crop_h, crop_w):
    """Window rows (y0, x0, flip) for ThreeCrop's oversampled group: crop-major, frame-minor, never mirrored (augmentations.py:512-530)."""
    return [(y0, x0, 0) for (x0, y0) in three_crop_offsets(img_h, img_w, crop_h, crop_w) for _ in range(n_frames)]


def flip_flag(flip_ratio, rng=None):
    """Flip's per-sample decision (augmentations.py:217): ONE `rand()` draw, mirrored when it is below flip_ratio."""
    import numpy as np
    return bool((rng if rng is not None else np.random).rand() < flip_ratio)


class FramePipeline(object):
    """Normalize(mean, std, to_rgb, div_255) + a crop size; per-frame windows (y0, x0, flip) select crop position and mirroring."""

    def __init__(self, mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375), to_rgb=True, div_255=False, crop_size=224):
        self.mean = (ctypes.c_float * 3)(*[float(v) for v in mean])
        self.std = (ctypes.c_float * 3)(*[float(v) for v in std])
        self.to_rgb, self.div_255 = bool(to_rgb), bool(div_255)
        self.crop_hw = (crop_size, crop_size) if isinstance(crop_size, int) else (int(crop_size[1]), int(crop_size[0]))   # cfg is (w, h)

    def center_window(self, n, hs, ws, flip=False, device="cuda"):
        """CenterCrop's window for every frame (augmentations.py:342-396: x0 = (W - w) // 2, y0 = (H - h) // 2)."""
        h, w = self.crop_hw
        row = [(hs - h) // 2, (ws - w) // 2, int(bool(flip))]
        return torch.tensor([row] * n, dtype=torch.int32, device=device)

    def _frames(self, frames):
        if frames.dtype != torch.uint8 or frames.shape[-1] != 3 or not frames.is_cuda:
            raise TypeError("FramePipeline expects a CUDA uint8 tensor (..., H, W, 3) of decoded frames, got %s %s" % (frames.dtype, tuple(frames.shape)))
        f = frames.reshape((-1,) + tuple(frames.shape[-3:])).contiguous()
        h, w = self.crop_hw
        if h > f.shape[1] or w > f.shape[2]:
            raise ValueError("crop %dx%d larger than the frames %dx%d" % (h, w, f.shape[1], f.shape[2]))
        return f

    def _window(self, window, n, hs, ws):
        if window is None:
            return None
        window = window.to(device="cuda", dtype=torch.int32).reshape(-1, 3).contiguous()
        if window.shape[0] != n:
            raise ValueError("window needs one (y0, x0, flip) row per frame: %d rows for %d frames" % (window.shape[0], n))
        h, w = self.crop_hw
        lo, hi = window.min(0).values.tolist(), window.max(0).values.tolist()
        if lo[0] < 0 or lo[1] < 0 or hi[0] + h > hs or hi[1] + w > ws:
            raise ValueError("crop window leaves the %dx%d frame" % (hs, ws))
        return window

    def to_nchw(self, frames, window=None):
        """-> (n, 3, h, w) fp32, what the reference's Normalize + FormatShape + ToTensor produce for these frames."""
        f = self._frames(frames)
        n, hs, ws = f.shape[:3]
        win = self._window(window, n, hs, ws)
        h, w = self.crop_hw
        out = torch.empty(n, 3, h, w, dtype=torch.float32, device=f.device)
        check(lib.mvf_frames_prep_u8(f.data_ptr(), n, hs, ws, win.data_ptr() if win is not None else None, h, w, self.mean, self.std,
                                     int(self.to_rgb), int(self.div_255), 0, w, None, out.data_ptr(), 0,
                                     torch.cuda.current_stream().cuda_stream), "mvf_frames_prep_u8")
        return out

    def to_stem(self, frames, window, pad, wp, dtype, out=None):
        """-> (n, h + 2 pad, wp, 4) `dtype`: the zero-padded channels-last operand of the 7x7 stem conv (what mvf_stem_prep makes
        from the fp32 NCHW tensor), straight from the uint8 frames."""
        f = self._frames(frames)
        n, hs, ws = f.shape[:3]
        win = self._window(window, n, hs, ws)
        h, w = self.crop_hw
        if out is None:
            out = torch.empty(n, h + 2 * pad, wp, 4, dtype=dtype, device=f.device)
        check(lib.mvf_frames_prep_u8(f.data_ptr(), n, hs, ws, win.data_ptr() if win is not None else None, h, w, self.mean, self.std,
                                     int(self.to_rgb), int(self.div_255), pad, wp, out.data_ptr(), None, _DT[dtype],
                                     torch.cuda.current_stream().cuda_stream), "mvf_frames_prep_u8")
        return out


# ---- frame-index arithmetic (host side; reference codes/datasets/pipelines/loading.py:11-131) ------------------------------------
def _train_offsets(total_frames, span, num_clips, rng):
    """SampleFrames._sample_clips (loading.py:35-60): start offsets of `num_clips` training clips of `span = clip_len * frame_interval`
    source frames.  Three cases, in the reference's order: equal segments with one random shift each; sorted random starts when the
    segments would be empty but the video is longer than max(num_clips, span); otherwise every clip starts at frame 0.
    Draws: ONE `randint(high, size=num_clips)` call in the first two cases, none in the third."""
    import numpy as np
    room = total_frames - span + 1
    seg = room // num_clips
    if seg > 0:
        return np.arange(num_clips, dtype=np.int64) * seg + np.asarray(rng.randint(seg, size=num_clips), dtype=np.int64)
    if total_frames > max(num_clips, span):
        return np.sort(np.asarray(rng.randint(room, size=num_clips), dtype=np.int64))
    return np.zeros(num_clips, dtype=np.int64)


def _test_offsets(total_frames, span, num_clips, sth_samples, rng):
    """SampleFrames._test_sample_clips (loading.py:62-94).  sth_samples 1: segment centres int(tick / 2 + tick * k) (zeros when
    tick <= 0); 2: the centres followed by the segment starts int(tick * k), no tick test; 10: ten training draws in a row; any other
    value s: the centres, then s - 1 rows of k * floor(room / num_clips) + randint(floor(room / num_clips))."""
    import numpy as np
    room = total_frames - span + 1
    tick = room / float(num_clips)
    centres = [int(tick / 2.0 + tick * k) for k in range(num_clips)]
    if sth_samples == 1:
        return np.array(centres, dtype=np.int64) if tick > 0 else np.zeros(num_clips, dtype=np.int64)
    if sth_samples == 2:
        return np.array(centres + [int(tick * k) for k in range(num_clips)], dtype=np.int64)
    if sth_samples == 10:
        return np.concatenate([_train_offsets(total_frames, span, num_clips, rng) for _ in range(10)])
    seg = room // num_clips                                   # the reference's `// float(num_clips)`: same value, as a float
    rows = [np.array(centres, dtype=np.int64)]
    for _ in range(sth_samples - 1):
        rows.append(np.arange(num_clips, dtype=np.int64) * seg + np.asarray(rng.randint(float(seg), size=num_clips), dtype=np.int64))
    return np.concatenate(rows)


def sample_frame_inds(total_frames, clip_len, frame_interval=1, num_clips=1, test_mode=False, temporal_jitter=False, sth_samples=1, rng=None):
    """The frame indices `SampleFrames(clip_len, frame_interval, num_clips, temporal_jitter, sth_samples)` writes to
    results['frame_inds'] (loading.py:96-116): clip-major, frame-minor int64 array of len(offsets) * clip_len entries --
    offset + frame * frame_interval (+ ONE jitter draw `randint(frame_interval, size=clip_len)` shared by all clips when
    temporal_jitter), clamped to total_frames - 1.  `rng`: numpy.random (default, the reference's global generator) or a RandomState."""
    import numpy as np
    rng = rng if rng is not None else np.random
    span = clip_len * frame_interval
    if test_mode:
        offs = _test_offsets(total_frames, span, num_clips, sth_samples, rng)
    else:
        offs = _train_offsets(total_frames, span, num_clips, rng)
    inds = offs[:, None] + np.arange(clip_len, dtype=np.int64)[None, :] * frame_interval
    if temporal_jitter:
        inds = inds + np.asarray(rng.randint(frame_interval, size=clip_len), dtype=np.int64)[None, :]
    return np.minimum(inds.reshape(-1), total_frames - 1).astype(np.int64)


class SampleFrames(object):
    """Pipeline step with the reference's name, constructor and result keys (loading.py:11-131); `total_frames` must be in `results`
    (the reference's fallback opens the video with mmcv.VideoReader: decoding stays on the host, out of this repo's scope)."""

    def __init__(self, clip_len, frame_interval=1, num_clips=1, temporal_jitter=False, sth_samples=1):
        self.clip_len, self.frame_interval, self.num_clips = clip_len, frame_interval, num_clips
        self.temporal_jitter, self.sth_samples = temporal_jitter, sth_samples

    def __call__(self, results):
        if "total_frames" not in results:
            raise KeyError("SampleFrames: results['total_frames'] is required (video probing is not part of this build)")
        results["frame_inds"] = sample_frame_inds(results["total_frames"], self.clip_len, self.frame_interval, self.num_clips,
                                                  bool(results["test_mode"]), self.temporal_jitter, self.sth_samples)
        results["clip_len"], results["frame_interval"] = self.clip_len, self.frame_interval
        results["num_clips"], results["sth_samples"] = self.num_clips, self.sth_samples
        return results


class Normalize(object):
    """Pipeline step with the reference's name, constructor and result keys (augmentations.py:343-376) over the device kernel: here
    `img_group` is a CUDA uint8 tensor (..., H, W, 3) of decoded frames and the result is the (n, 3, H, W) fp32 tensor the reference's
    Normalize + FormatShape('NCHW') produce -- float32(img) [/ 255] -> channel swap when to_rgb -> subtract float32(mean) -> multiply by
    float32(1 / float64(std)), each a single rounded fp32 operation (mvf_frames_prep_u8)."""

    def __init__(self, mean, std, div_255=False, to_rgb=False):
        import numpy as np
        self.mean, self.std = np.array(mean, dtype=np.float32), np.array(std, dtype=np.float32)
        self.div_255, self.to_rgb = div_255, to_rgb

    def __call__(self, results):
        f = results["img_group"]
        pipe = FramePipeline(self.mean.tolist(), self.std.tolist(), to_rgb=self.to_rgb, div_255=self.div_255, crop_size=(f.shape[-2], f.shape[-3]))
        results["img_group"] = pipe.to_nchw(f)
        results["img_norm_cfg"] = dict(mean=self.mean, std=self.std, div_255=self.div_255, to_rgb=self.to_rgb)
        return results
