"""Input-pipeline oracle (oracle/frames_numpy.py): FormatShape against the reference's own output, crop offsets by formula."""
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
