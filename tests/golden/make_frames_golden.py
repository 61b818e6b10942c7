#!/usr/bin/env python3
"""Golden vector for the FormatShape step of the input pipeline, produced by the REFERENCE's own class
(codes/datasets/pipelines/formating.py:133-185), imported with the inert mmcv / cv2 placeholders of make_golden.py.
Run in the build container: python tests/golden/make_frames_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

mg._install_stubs()
sys.path.insert(0, mg.REF)
from codes.datasets.pipelines.formating import FormatShape  # noqa: E402

rng = np.random.RandomState(7)
frames = [rng.randn(9, 11, 3).astype(np.float32) for _ in range(6)]          # 6 HWC frames as Normalize leaves them
res = FormatShape("NCHW")(dict(img_group=list(frames), modality="RGB", num_clips=3, clip_len=2))
np.savez_compressed(os.path.join(HERE, "frames_cases.npz"), frames=np.stack(frames), nchw=res["img_group"],
                    input_shape=np.array(res["input_shape"]))
print("wrote frames_cases.npz", res["img_group"].shape)
