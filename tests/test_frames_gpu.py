"""mvf_frames_prep_u8 (uint8 frames -> crop / flip / normalise / channels-first / stem layout) against oracle/frames_numpy.py,
and the engines fed with uint8 frames against the same engines fed with the oracle's fp32 tensor."""
import numpy as np
import pytest
import torch

from oracle import frames_numpy as F

pytestmark = pytest.mark.gpu

MEAN, STD = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]


def _frames(n, hs, ws, seed):
    rng = np.random.RandomState(seed)
    return rng.randint(0, 256, size=(n, hs, ws, 3)).astype(np.uint8)


@pytest.mark.parametrize("case", [
    (5, 37, 53, 21, 32, True, False),      # odd sizes, every frame its own window, some mirrored
    (3, 64, 64, 64, 64, False, False),     # no crop, BGR kept
    (4, 40, 48, 33, 17, True, True),       # div_255
    (2, 9, 9, 1, 1, True, False),          # 1x1 crop
], ids=["odd_windows", "full_frame_bgr", "div255", "one_pixel"])
def test_frames_prep_nchw_bit_exact_vs_oracle(case):
    from mvfnet_amd.preprocess import FramePipeline
    n, hs, ws, h, w, to_rgb, div = case
    fr = _frames(n, hs, ws, 11)
    rng = np.random.RandomState(5)
    win = np.stack([rng.randint(0, hs - h + 1, n), rng.randint(0, ws - w + 1, n), rng.randint(0, 2, n)], 1).astype(np.int32)
    mean, std = (MEAN, STD) if not div else ([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    pipe = FramePipeline(mean, std, to_rgb=to_rgb, div_255=div, crop_size=(w, h))
    got = pipe.to_nchw(torch.from_numpy(fr).cuda(), torch.from_numpy(win).cuda()).cpu().numpy()
    want = F.frames_to_nchw(fr, win, h, w, mean, std, to_rgb=to_rgb, div_255=div)
    assert np.array_equal(got, want)                       # subtraction and multiplication are single rounded fp32 steps: bit exact
    got0 = pipe.to_nchw(torch.from_numpy(fr).cuda(), None).cpu().numpy()
    assert np.array_equal(got0, F.frames_to_nchw(fr, None, h, w, mean, std, to_rgb=to_rgb, div_255=div))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_frames_prep_stem_layout_equals_stem_prep_of_the_oracle_tensor(dtype):
    """The fused kernel's stem operand == mvf_stem_prep applied to the oracle's normalised tensor (same padding, same rounding)."""
    from mvfnet_amd._lib import check, lib
    from mvfnet_amd.preprocess import FramePipeline
    n, hs, ws, h, w, pad = 3, 50, 61, 40, 44, 3
    fr = _frames(n, hs, ws, 3)
    win = np.array([[0, 0, 0], [10, 17, 1], [5, 3, 1]], dtype=np.int32)
    pipe = FramePipeline(MEAN, STD, to_rgb=True, crop_size=(w, h))
    wp = (w + 2 * pad + 2 + 1) // 2 * 2
    got = pipe.to_stem(torch.from_numpy(fr).cuda(), torch.from_numpy(win).cuda(), pad, wp, dtype)
    x = torch.from_numpy(F.frames_to_nchw(fr, win, h, w, MEAN, STD, to_rgb=True)).cuda()
    ref = torch.full((n, h + 2 * pad, wp, 4), 7.0, dtype=dtype, device="cuda")
    check(lib.mvf_stem_prep(x.data_ptr(), n, 3, h, w, pad, wp, ref.data_ptr(), 0 if dtype == torch.float32 else 1,
                            torch.cuda.current_stream().cuda_stream), "stem_prep")
    assert torch.equal(got.view(torch.int16 if dtype == torch.bfloat16 else torch.int32),
                       ref.view(torch.int16 if dtype == torch.bfloat16 else torch.int32))


def test_frames_prep_rejects_bad_input():
    from mvfnet_amd.preprocess import FramePipeline
    pipe = FramePipeline(MEAN, STD, crop_size=16)
    fr = torch.zeros(2, 20, 20, 3, dtype=torch.uint8, device="cuda")
    with pytest.raises(TypeError):
        pipe.to_nchw(fr.float())
    with pytest.raises(ValueError):
        pipe.to_nchw(fr, torch.tensor([[5, 0, 0], [0, 0, 0]], dtype=torch.int32))        # y0 + 16 > 20
    with pytest.raises(ValueError):
        pipe.to_nchw(fr, torch.tensor([[0, 0, 0]], dtype=torch.int32))                   # one row for two frames
    with pytest.raises(ValueError):
        FramePipeline(MEAN, STD, crop_size=32).to_nchw(fr)
    with pytest.raises(RuntimeError):
        FramePipeline(MEAN, [1.0, 0.0, 1.0], crop_size=16).to_nchw(fr)                   # std 0 -> status code from the library


def _r50(T):
    import mvfnet_amd
    from mvfnet_amd import synth
    m = mvfnet_amd.build_recognizer(mvfnet_amd.mvfnet_config(50, T), None, dict(average_clips=None))
    sd = m.state_dict()
    vals = synth.synth_state_dict({"r50/" + k: tuple(v.shape) for k, v in sd.items()})
    m.load_state_dict({k: torch.from_numpy(vals["r50/" + k]) for k in sd}, strict=True)
    return m.cuda()


def test_engines_take_uint8_frames_and_match_the_fp32_tensor_path():
    from mvfnet_amd.preprocess import FramePipeline
    T, B, hs, ws, c = 4, 2, 72, 80, 64
    m = _r50(T)
    fr = _frames(B * T, hs, ws, 9)
    win = np.stack([np.full(B * T, 3), np.full(B * T, 7), np.repeat([0, 1], T)], 1).astype(np.int32)     # clip 1 mirrored
    pipe = FramePipeline(MEAN, STD, to_rgb=True, crop_size=c)
    x = torch.from_numpy(F.frames_to_nchw(fr, win, c, c, MEAN, STD, to_rgb=True)).cuda()
    labels = torch.tensor([[3], [111]], device="cuda")
    # inference engine
    m.eval()
    eng = m.backbone.engine()
    want = eng.forward(x).float().clone()
    eng.input_pipeline, eng.input_window = pipe, torch.from_numpy(win).cuda()
    got = eng.forward(torch.from_numpy(fr).cuda()).float()
    assert torch.equal(got, want)
    # train engine: same stem operand -> same loss, up to the fp32 rounding of the batch statistics (their summation shift is
    # the BN's running mean, which the first forward has moved)
    m.train()
    te = m.train_engine()
    te.dropout = 0.0
    l0 = float(te.forward(x.view(B, T, 3, c, c), labels))
    te.input_pipeline, te.input_window = pipe, torch.from_numpy(win).cuda()
    l1 = float(te.forward(torch.from_numpy(fr).cuda().view(B, T, hs, ws, 3), labels))
    assert l1 == pytest.approx(l0, rel=1e-5)


def test_recognizer_api_with_uint8_frames_matches_fp32_input():
    """model(img_group=uint8 frames, window=...) == model(img_group=the oracle's normalised tensor) through the public API,
    test mode (scores) and train mode (loss_cls)."""
    from mvfnet_amd.preprocess import FramePipeline
    T, B, hs, ws, c = 4, 2, 70, 90, 64
    m = _r50(T)
    fr = _frames(B * T, hs, ws, 21)
    win = np.stack([np.full(B * T, 2), np.full(B * T, 11), np.repeat([1, 0], T)], 1).astype(np.int32)
    x = torch.from_numpy(F.frames_to_nchw(fr, win, c, c, MEAN, STD, to_rgb=True)).cuda().view(B, T, 3, c, c)
    fr_t, win_t = torch.from_numpy(fr).cuda().view(B, T, hs, ws, 3), torch.from_numpy(win).cuda()
    m.eval()
    want = m(x, None, return_loss=False)
    m.set_input_pipeline(FramePipeline(MEAN, STD, to_rgb=True, crop_size=c))
    got = m(fr_t, None, return_loss=False, window=win_t)
    assert np.array_equal(got, want)
    m.train()
    m.cls_head.dropout = None
    lab = torch.tensor([[5], [77]], device="cuda")
    l0 = float(m(x, lab)["loss_cls"].detach())
    l1 = float(m(fr_t, lab, window=win_t)["loss_cls"].detach())
    assert l1 == pytest.approx(l0, rel=1e-5)


def test_normalize_kernel_bit_exact_vs_the_reference_class_golden():
    """[r4] mvf_frames_prep_u8 (and the Normalize pipeline step over it) against the images the REFERENCE's Normalize class returned
    (tests/golden/normalize_cases.npz, make_normalize_golden.py): bit-equal for K400's mean / std with and without to_rgb, the div_255
    convention, and every uint8 value in every channel."""
    import os
    from mvfnet_amd.preprocess import FramePipeline, Normalize
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "normalize_cases.npz"))
    for tag in sorted({k.split("/")[0] for k in G.files}):
        fr = G[tag + "/frames"]
        div, rgb = (bool(v) for v in G[tag + "/cfg_flags"])
        mean, std = G[tag + "/cfg_mean"], G[tag + "/cfg_std"]
        want = G[tag + "/out"].transpose(0, 3, 1, 2)
        pipe = FramePipeline(mean.tolist(), std.tolist(), to_rgb=rgb, div_255=div, crop_size=(fr.shape[2], fr.shape[1]))
        got = pipe.to_nchw(torch.from_numpy(fr).cuda(), None).cpu().numpy()
        assert np.array_equal(got, want), tag
        res = Normalize(mean, std, div_255=div, to_rgb=rgb)(dict(img_group=torch.from_numpy(fr).cuda()))
        assert np.array_equal(res["img_group"].cpu().numpy(), want), tag
        assert np.array_equal(res["img_norm_cfg"]["mean"], mean) and res["img_norm_cfg"]["to_rgb"] == rgb
