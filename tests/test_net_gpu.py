"""GPU: the whole eval-mode hot path (Recognizer2D.forward_test through the HIP engine) against the golden vectors
captured from the reference and against the CPU oracle."""
import numpy as np
import pytest
import torch

from helpers import golden, rel_err
from mvfnet_amd import synth

pytestmark = pytest.mark.gpu

TOL_F32 = 1e-4    # relative to each tensor's max; north_star budget 1e-3 (fp32)
TOL_BF16 = 1e-2   # north_star: 1e-2 bf16, relative to the output scale (measured ~2e-3 on the logits; see also test_bf16_parity_gpu.py)


def _model(depth, T, average_clips=None, dtype=torch.float32, fcn=False):
    import mvfnet_amd
    m = mvfnet_amd.build_recognizer(mvfnet_amd.mvfnet_config(depth, T, fcn_testing=fcn), None, dict(average_clips=average_clips))
    sd = m.state_dict()
    pre = "r%d/" % depth
    vals = synth.synth_state_dict({pre + k: tuple(v.shape) for k, v in sd.items()})
    m.load_state_dict({k: torch.from_numpy(vals[pre + k]) for k in sd}, strict=True)
    m.backbone.engine_dtype = dtype
    return m.cuda().eval()


def _stage_check(stages, g, prefix, tol):
    for k, v in stages.items():
        a = v.float().cpu().permute(0, 3, 1, 2).contiguous().numpy().astype(np.float64).ravel()   # NHWC buffer -> NCHW order
        ref = g[prefix + k]
        assert abs(a.mean() - ref[0]) < tol * ref[2], k
        assert abs(np.sqrt((a * a).mean()) - ref[1]) < tol * ref[2], k
        idx = np.linspace(0, a.size - 1, 16).astype(np.int64)
        assert np.abs(a[idx] - ref[3:]).max() < tol * ref[2], k


def test_c1_r50_4x16_eval_logits_and_stages():
    """BASELINE config 1 (R50 4x16, 2 clips 224^2) on the GPU: logits, per-stage checksums, top-1, clip averaging."""
    g = golden("net_cases.npz")
    m = _model(50, 4)
    imgs = torch.from_numpy(synth.synth_clip_batch(2, 4, 224, 224)).cuda()
    stages = {}
    x = imgs.reshape(-1, 3, 224, 224)
    feat = m.backbone(x, stages=stages)
    _stage_check(stages, g, "c1/eval/stage/", TOL_F32)
    logits = m(imgs, None, return_loss=False, return_numpy=True)
    assert logits.shape == (2, 400)
    assert rel_err(logits, g["c1/eval/logits"]) < TOL_F32
    assert (logits.argmax(1) == g["c1/eval/logits"].argmax(1)).all()       # class-index exact
    m.test_cfg = dict(average_clips="prob")
    assert rel_err(m(imgs, None, return_loss=False), g["c1/eval/prob"]) < TOL_F32
    m.test_cfg = dict(average_clips="score")
    assert rel_err(m(imgs, None, return_loss=False), g["c1/eval/score"]) < TOL_F32


@pytest.mark.parametrize("depth,T,tag", [(50, 8, "r50_t8"), (101, 16, "r101_t16")])
def test_r50_8x8_and_r101_16x4_logits(depth, T, tag):
    g = golden("net_cases.npz")
    m = _model(depth, T)
    imgs = torch.from_numpy(synth.synth_clip_batch(1, T, 224, 224, seed=depth)).cuda()
    logits = m(imgs, None, return_loss=False)
    assert rel_err(logits, g[tag + "/eval/logits"]) < TOL_F32
    assert (logits.argmax(1) == g[tag + "/eval/logits"].argmax(1)).all()


def test_fcn_testing_video():
    """config-5 style: 1 video = 3 crops x 2 clips x T frames (128^2), fcn_testing head, average_clips='prob'."""
    g = golden("net_cases.npz")
    m = _model(50, 4, "prob", fcn=True)
    vid = torch.from_numpy(synth.synth_tensor("fcn_video", (1, 3 * 2 * 4, 3, 128, 128))).cuda()
    prob = m(vid, None, return_loss=False)
    assert prob.shape == (1, 400)
    assert rel_err(prob, g["fcn/prob"]) < TOL_F32
    assert prob.argmax() == g["fcn/prob"].argmax()
    m.test_cfg = dict(average_clips=None)
    assert rel_err(m(vid, None, return_loss=False), g["fcn/scores"]) < TOL_F32


def test_c1_bf16_engine_within_budget():
    g = golden("net_cases.npz")
    m = _model(50, 4, dtype=torch.bfloat16)
    imgs = torch.from_numpy(synth.synth_clip_batch(2, 4, 224, 224)).cuda()
    logits = m(imgs, None, return_loss=False)
    assert rel_err(logits, g["c1/eval/logits"]) < TOL_BF16
    assert (logits.argmax(1) == g["c1/eval/logits"].argmax(1)).all()


def test_c2_full_batch_clip_independence_and_oracle():
    """BASELINE config 2 size (32 clips x 8 frames, 224^2, fp32): clips are independent units in eval mode, so the
    batch-32 output must reproduce (a) the golden N=1 logits for clip 0 and (b) any permutation of clips."""
    g = golden("net_cases.npz")
    m = _model(50, 8)
    one = torch.from_numpy(synth.synth_clip_batch(1, 8, 224, 224, seed=50)).cuda()
    rest = torch.randn(31, 8, 3, 224, 224, device="cuda", generator=torch.Generator("cuda").manual_seed(5))
    batch = torch.cat([one, rest], 0)
    out = m(batch, None, return_loss=False, return_numpy=False)
    assert out.shape == (32, 400)
    assert rel_err(out[:1].cpu().numpy(), g["r50_t8/eval/logits"]) < TOL_F32
    perm = torch.randperm(32, device="cuda")
    outp = m(batch[perm], None, return_loss=False, return_numpy=False)
    assert rel_err(outp.cpu().numpy(), out[perm].cpu().numpy()) < 1e-5
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_c1_logits_with_mvf_fused_into_the_conv_loader(dtype, monkeypatch):
    """The opt-in single-launch MVF + conv1 (mvf_conv2d_nhwc_fwd_mvf) through the whole network: the reference's logits again."""
    from mvfnet_amd import engine
    monkeypatch.setattr(engine, "FUSE_MVF_LOADER", True)
    g = golden("net_cases.npz")
    m = _model(50, 4, dtype=dtype)
    assert all(b.fuse_mvf for b in m.backbone.engine().blocks if b.mvf is not None)
    imgs = torch.from_numpy(synth.synth_clip_batch(2, 4, 224, 224)).cuda()
    logits = m(imgs, None, return_loss=False)
    assert rel_err(logits, g["c1/eval/logits"]) < (TOL_F32 if dtype == torch.float32 else TOL_BF16)
    assert (logits.argmax(1) == g["c1/eval/logits"].argmax(1)).all()
