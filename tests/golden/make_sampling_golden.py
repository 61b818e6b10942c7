#!/usr/bin/env python3
"""Golden vectors for the frame-index arithmetic of the input pipeline, produced by the REFERENCE's own class
(codes/datasets/pipelines/loading.py:11-131: SampleFrames._sample_clips :35-60, ._test_sample_clips :62-94, ._get_frame_inds :96-116,
.__call__ :118-135) imported with the mmcv / cv2 placeholders of make_golden.py.  The only shim is `np.int = int` (the alias the
reference's `.astype(np.int)` at loading.py:115 needs; removed from numpy 1.24 on).

Stored arrays = data only: for every case the constructor / call arguments, the np.random seed, and the `frame_inds` the reference
returned (+ the other keys __call__ writes).  Cases cover
  * the three branches of _sample_clips (avg_interval > 0; num_frames > max(num_clips, ori_clip_len); the all-zero fallback),
  * _test_sample_clips for sth_samples 1 and 2 (tick > 0 and tick <= 0), 10 (ten random draws) and the generic branch (3),
  * temporal_jitter (one draw of clip_len offsets shared by all clips) and the `minimum(total_frames - 1)` clamp,
  * the shipped recipes: training 1 clip x 8 frames x interval 8 (R50 8x8), 16 x 4 (R101), and the C5 test recipe 10 clips x 8 x 8.

Run in the build container: python tests/golden/make_sampling_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

mg._install_stubs()
if not hasattr(np, "int"):
    np.int = int          # loading.py:115 `.astype(np.int)`

sys.path.insert(0, mg.REF)
from codes.datasets.pipelines.loading import SampleFrames  # noqa: E402

# (tag, total_frames, clip_len, frame_interval, num_clips, test_mode, temporal_jitter, sth_samples)
CASES = [
    ("train_r50_8x8_300", 300, 8, 8, 1, False, False, 1),
    ("train_r50_8x8_250", 250, 8, 8, 1, False, False, 1),
    ("train_r101_16x4_300", 300, 16, 4, 1, False, False, 1),
    ("train_4x16_120", 120, 4, 16, 1, False, False, 1),
    ("train_branch1_3clips", 100, 4, 2, 3, False, False, 1),
    ("train_branch1_jitter", 100, 4, 3, 3, False, True, 1),
    ("train_branch2_sorted", 12, 4, 2, 5, False, False, 1),       # avg_interval = (12 - 8 + 1) // 5 = 1 > 0 -> branch 1
    ("train_branch2_real", 10, 4, 2, 5, False, False, 1),         # (10 - 8 + 1) // 5 = 0, 10 > max(5, 8): sorted random offsets
    ("train_branch2_jitter_clamp", 10, 4, 2, 5, False, True, 1),  # jitter can push past the end: clamp to total_frames - 1
    ("train_branch3_zeros", 7, 4, 2, 5, False, False, 1),         # 7 <= max(5, 8): all-zero offsets (float array in the reference)
    ("train_branch3_short_video", 5, 8, 8, 1, False, False, 1),   # clip longer than the video: zeros + clamp
    ("train_branch3_jitter", 6, 4, 2, 3, False, True, 1),
    ("test_c5_10x8x8_300", 300, 8, 8, 10, True, False, 1),        # BASELINE configs[4]: 10 clips x 8 frames x interval 8
    ("test_c5_10x8x8_250", 250, 8, 8, 10, True, False, 1),
    ("test_c5_10x8x8_64", 64, 8, 8, 10, True, False, 1),          # tick = 0.1
    ("test_c5_10x8x8_63", 63, 8, 8, 10, True, False, 1),          # tick = 0 -> zeros
    ("test_c5_10x8x8_40", 40, 8, 8, 10, True, False, 1),          # tick < 0 -> zeros + clamp
    ("test_1clip_4x16", 150, 4, 16, 1, True, False, 1),
    ("test_sth2", 80, 8, 2, 2, True, False, 2),
    ("test_sth2_short", 12, 8, 2, 2, True, False, 2),             # negative tick: int() truncation towards zero, clamp below? (indices stay >= 0 here)
    ("test_sth10", 90, 4, 2, 2, True, False, 10),
    ("test_sth3_generic", 90, 4, 2, 3, True, False, 3),
    ("test_jitter", 100, 4, 3, 2, True, True, 1),
]
SEEDS = (0, 1, 7)

out = {}
names = []
for (tag, total, clip_len, interval, num_clips, test_mode, jitter, sth) in CASES:
    for seed in SEEDS:
        key = "%s/seed%d" % (tag, seed)
        np.random.seed(seed)
        sf = SampleFrames(clip_len, interval, num_clips, jitter, sth)
        try:
            res = sf(dict(total_frames=total, test_mode=test_mode))
        except Exception as e:      # a branch the installed numpy refuses (float `high` of randint): recorded as absent
            print("skipped %s: %s" % (key, str(e)[:100]))
            continue
        out[key + "/args"] = np.array([total, clip_len, interval, num_clips, int(test_mode), int(jitter), sth, seed], dtype=np.int64)
        out[key + "/frame_inds"] = np.asarray(res["frame_inds"])
        assert res["frame_inds"].dtype == np.int64
        out[key + "/keys"] = np.array([res["clip_len"], res["frame_interval"], res["num_clips"], res["sth_samples"]], dtype=np.int64)
        # the generator state after the call pins HOW MANY draws the reference made
        out[key + "/next_draw"] = np.array(np.random.randint(1 << 30))
        names.append(key)
np.savez_compressed(os.path.join(HERE, "sampling_cases.npz"), **out)
print("wrote sampling_cases.npz: %d arrays, %d cases" % (len(out), len(names)))
