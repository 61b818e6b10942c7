"""Input-pipeline oracle (oracle/frames_numpy.py): FormatShape against the reference's own output; [r3] the crop / flip decisions against
what the reference's ThreeCrop / CenterCrop / Flip handed to (recording stand-ins of) mmcv.imcrop / imflip (tests/golden/make_crops_golden.py)."""
import os

import numpy as np

from oracle import frames_numpy as F

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "frames_cases.npz"))


def test_format_shape_nchw_matches_reference_golden():
    frames = G["frames"]                       # (6, 9, 11, 3) float32 HWC, as Normalize leaves them
    got = np.stack([f.transpose(2, 0, 1) for f in frames], axis=0)
    assert tuple(G["input_shape"]) == got.shape
    assert np.array_equal(got, G["nchw"])
    # the oracle's own stacking is that transpose (identity normalisation: mean 0, std 1, no channel swap)
    u8 = (np.arange(2 * 5 * 7 * 3) % 251).astype(np.uint8).reshape(2, 5, 7, 3)
    out = F.frames_to_nchw(u8, None, 5, 7, [0, 0, 0], [1, 1, 1], to_rgb=False)
    assert np.array_equal(out, np.float32(u8).transpose(0, 3, 1, 2))


def test_three_crop_offsets_follow_the_reference_cases():
    # crop_h == img_h (the fcn_testing 256x256 crops of a 256 x 340 frame): left, right, middle along the width
    assert F.three_crop_offsets(256, 340, 256, 256) == [(0, 0), (84, 0), (42, 0)]
    assert F.three_crop_offsets(340, 256, 256, 256) == [(0, 0), (0, 84), (0, 42)]
    assert F.three_crop_offsets(300, 400, 224, 224) == [(0, 38), (176, 38), (88, 38)]
    from mvfnet_amd.preprocess import three_crop_offsets
    for a in [(256, 340, 256, 256), (340, 256, 256, 256), (300, 400, 224, 224)]:
        assert three_crop_offsets(*a) == F.three_crop_offsets(*a)


def test_normalize_is_two_rounded_fp32_steps():
    u8 = np.arange(256, dtype=np.uint8).reshape(1, 16, 16, 1).repeat(3, axis=3)
    mean, std = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
    out = F.frames_to_nchw(u8, None, 16, 16, mean, std, to_rgb=True)
    m32, s32 = np.float32(mean), np.float32(std)
    for k in range(3):                           # to_rgb: output channel k reads input channel 2-k (equal here), uses mean[k]
        want = np.float32(np.float32(np.float32(u8[0, :, :, 2 - k]) - m32[k]) * np.float32(1.0 / np.float64(s32[k])))
        assert np.array_equal(out[0, k], want)


# ------------------------------------------------------------------------------------------------ [r3] crop / flip decisions vs the reference's classes
GC = np.load(os.path.join(os.path.dirname(__file__), "golden", "crops_cases.npz"))


def _three_cases():
    return sorted({k.split("/")[1] for k in GC.files if k.startswith("three_crop/")})


def test_three_crop_boxes_and_group_order_match_the_reference():
    from mvfnet_amd import preprocess as P
    assert len(_three_cases()) == 8
    for case in _three_cases():
        hw, crop = case.split("_")
        h, w = (int(v) for v in hw.split("x"))
        cw, ch = (int(v) for v in crop.split("x"))
        boxes, ret = GC["three_crop/%s/boxes" % case], GC["three_crop/%s/returned" % case]
        n = boxes.shape[0] // 3
        for mod in (F, P):
            offs = mod.three_crop_offsets(h, w, ch, cw)
            # the boxes the reference hands to mmcv.imcrop, call by call: [x0, y0, x0 + w - 1, y0 + h - 1], every offset for all frames
            want = np.array([[x0, y0, x0 + cw - 1, y0 + ch - 1] for (x0, y0) in offs for _ in range(n)])
            assert np.array_equal(boxes, want), (case, mod.__name__)
            win = np.asarray(mod.three_crop_windows(n, h, w, ch, cw))
            # what comes back, image by image: frame index (frame-minor), top-left corner, crop size
            assert np.array_equal(ret[:, 0], np.tile(np.arange(n), 3)), case
            assert np.array_equal(ret[:, 1:3], win[:, :2]) and not win[:, 2].any(), (case, mod.__name__)
            assert np.array_equal(ret[:, 3:], np.tile([ch, cw], (3 * n, 1)))
        assert tuple(GC["three_crop/%s/img_shape" % case]) == (ch, cw, 3)
        # the oracle's crop semantics on marked frames = the windows above (frames_to_nchw slices [y0:y0+h, x0:x0+w])
        marks = np.zeros((n, h, w, 3), dtype=np.uint8)
        marks[..., 1] = (np.arange(h) % 251)[None, :, None]
        marks[..., 2] = (np.arange(w) % 241)[None, None, :]
        win = F.three_crop_windows(n, h, w, ch, cw)
        out = F.frames_to_nchw(np.concatenate([marks] * 3), win, ch, cw, [0, 0, 0], [1, 1, 1], to_rgb=False)
        assert np.array_equal(out[:, 1, 0, 0], ret[:, 1] % 251) and np.array_equal(out[:, 2, 0, 0], ret[:, 2] % 241)


def test_center_crop_box_matches_the_reference():
    import torch
    from mvfnet_amd.preprocess import FramePipeline
    cases = sorted({k.split("/")[1] for k in GC.files if k.startswith("center_crop/")})
    assert len(cases) == 3
    for case in cases:
        hw, crop = case.split("_")
        h, w = (int(v) for v in hw.split("x"))
        cw, ch = (int(v) for v in crop.split("x"))
        box = GC["center_crop/%s/box" % case]
        assert np.array_equal(GC["center_crop/%s/logged" % case], np.tile(box, (2, 1)))
        x0, y0 = F.center_crop_offset(h, w, ch, cw)
        assert box.tolist() == [x0, y0, x0 + cw - 1, y0 + ch - 1]
        assert np.array_equal(GC["center_crop/%s/returned" % case][:, 1:], np.tile([y0, x0, ch, cw], (2, 1)))
        row = FramePipeline(crop_size=(cw, ch)).center_window(2, h, w, device="cpu")      # cfg order (w, h), as the reference's crop_size
        assert row.tolist() == [[y0, x0, 0]] * 2 and row.dtype == torch.int32


def test_flip_decision_is_one_draw_below_the_ratio():
    from mvfnet_amd.preprocess import flip_flag
    seen = set()
    for seed in range(8):
        for ratio in (0.5, 0.0, 1.0):
            tag = "flip/seed%d_ratio%g" % (seed, ratio)
            flag = int(GC[tag + "/flag"])
            assert int(GC[tag + "/calls"]) == 2 * flag                   # both frames mirrored, or none
            assert int(GC[tag + "/first_col"]) == (5 if flag else 0)
            draw = np.random.RandomState(seed).rand()                   # = np.random.seed(seed); np.random.rand()
            assert F.flip_decision(draw, ratio) == bool(flag), tag
            assert flip_flag(ratio, np.random.RandomState(seed)) == bool(flag), tag
            seen.add((ratio, flag))
    assert {(0.5, 0), (0.5, 1), (0.0, 0), (1.0, 1)} <= seen


# ------------------------------------------------------------------------------------------------ [r4] SampleFrames / Normalize vs the reference's classes
GS = np.load(os.path.join(os.path.dirname(__file__), "golden", "sampling_cases.npz"))
GN = np.load(os.path.join(os.path.dirname(__file__), "golden", "normalize_cases.npz"))


def _sampling_cases():
    return sorted({k.rsplit("/", 1)[0] for k in GS.files})


def test_sample_frame_inds_match_the_reference_sampleframes():
    """oracle/frames_numpy.sample_frame_inds AND mvfnet_amd.preprocess.sample_frame_inds / SampleFrames against what the reference's
    SampleFrames returned (tests/golden/make_sampling_golden.py): indices bit-equal, and the same NUMBER of random draws (the next
    draw of the seeded generator after the call is the recorded one)."""
    from mvfnet_amd import preprocess as P
    cases = _sampling_cases()
    assert len(cases) == 69
    branches = set()
    for c in cases:
        total, clip_len, interval, num_clips, test_mode, jitter, sth, seed = (int(v) for v in GS[c + "/args"])
        want = GS[c + "/frame_inds"]
        assert want.dtype == np.int64 and want.max() <= total - 1      # (sth_samples = 2 on a too-short video yields NEGATIVE indices in the reference: kept)
        for fn in (F.sample_frame_inds, P.sample_frame_inds):
            np.random.seed(seed)
            got = fn(total, clip_len, interval, num_clips, bool(test_mode), bool(jitter), sth)
            assert got.dtype == np.int64 and np.array_equal(got, want), (c, fn.__module__)
            assert np.random.randint(1 << 30) == int(GS[c + "/next_draw"]), (c, "number of draws")
        # the pipeline-step form writes the same keys as the reference's __call__
        np.random.seed(seed)
        res = P.SampleFrames(clip_len, interval, num_clips, bool(jitter), sth)(dict(total_frames=total, test_mode=bool(test_mode)))
        assert np.array_equal(res["frame_inds"], want)
        assert [res["clip_len"], res["frame_interval"], res["num_clips"], res["sth_samples"]] == GS[c + "/keys"].tolist()
        # an explicit RandomState gives the same stream as the seeded global generator
        got = P.sample_frame_inds(total, clip_len, interval, num_clips, bool(test_mode), bool(jitter), sth, rng=np.random.RandomState(seed))
        assert np.array_equal(got, want)
        span = clip_len * interval
        if not test_mode:
            branches.add(1 if (total - span + 1) // num_clips > 0 else (2 if total > max(num_clips, span) else 3))
    assert branches == {1, 2, 3}                  # every branch of _sample_clips is in the fixture
    # the C5 recipe: 10 clips x 8 frames x interval 8 of a 300-frame video -> 80 indices, clip-major
    c5 = GS["test_c5_10x8x8_300/seed0/frame_inds"]
    assert c5.shape == (80,) and np.array_equal(np.diff(c5.reshape(10, 8), axis=1), np.full((10, 7), 8))


def test_normalize_matches_the_reference_class_call_by_call():
    """The reference's Normalize (augmentations.py:343-386) run over recording stand-ins of the three cv2 primitives
    (tests/golden/make_normalize_golden.py): the recorded call order / operands are what oracle/frames_numpy.imnormalize restates,
    and its output equals the class's output bit for bit."""
    tags = sorted({k.split("/")[0] for k in GN.files})
    assert tags == ["all_values", "div255_bgr", "div255_rgb", "k400_bgr", "k400_rgb"]
    for tag in tags:
        fr = GN[tag + "/frames"]
        div, rgb = (bool(v) for v in GN[tag + "/cfg_flags"])
        mean, std = GN[tag + "/cfg_mean"], GN[tag + "/cfg_std"]
        assert mean.dtype == np.float32 and std.dtype == np.float32          # Normalize.__init__: np.array(..., dtype=np.float32)
        per_img = [0, 1, 2] if rgb else [1, 2]                               # cvtColor BEFORE subtract BEFORE multiply, per image
        assert GN[tag + "/calls"].tolist() == per_img * fr.shape[0]
        assert int(GN[tag + "/inplace"]) == 1 and int(GN[tag + "/scalars_constant"]) == 1
        # operands: float64 (1, 3): float64(float32(mean)), 1 / float64(float32(std))
        assert GN[tag + "/sub_scalar"].dtype == np.float64 and GN[tag + "/sub_scalar"].shape == (1, 3)
        assert np.array_equal(GN[tag + "/sub_scalar"], np.float64(mean.reshape(1, -1)))
        assert np.array_equal(GN[tag + "/mul_scalar"], 1 / np.float64(std.reshape(1, -1)))
        # div_255: the image enters imnormalize as float32 (uint8 / 255 in fp32), otherwise as uint8
        assert GN[tag + "/in_dtype"].tolist() == [1 if div else 0] * fr.shape[0]
        got = np.stack([F.imnormalize(f, mean, std, rgb, div) for f in fr])
        assert got.dtype == np.float32 and np.array_equal(got, GN[tag + "/out"]), tag
        # and through the stacking step
        nchw = F.frames_to_nchw(fr, None, fr.shape[1], fr.shape[2], mean, std, to_rgb=rgb, div_255=div)
        assert np.array_equal(nchw, GN[tag + "/out"].transpose(0, 3, 1, 2))
