"""CPU: checkpoint wire format, lr schedule and loss parsing of the runner shell."""
import os

import pytest
import torch


def test_step_lr_with_linear_warmup_matches_mmcv_formula():
    from mvfnet_amd.runner import step_lr
    assert abs(step_lr(0.015, 0, 0) - 0.015 * 0.01) < 1e-12                       # warmup_ratio at iter 0
    assert abs(step_lr(0.015, 0, 25070) - 0.015) < 1e-12                          # warm-up over
    t = 10000
    assert abs(step_lr(0.015, 0, t) - 0.015 * (1 - (1 - t / 25070.0) * 0.99)) < 1e-12
    assert abs(step_lr(0.015, 90, 10 ** 6) - 0.0015) < 1e-12 and abs(step_lr(0.015, 130, 10 ** 6) - 0.00015) < 1e-12


def test_parse_losses():
    from mvfnet_amd.runner import parse_losses
    loss, logs = parse_losses(dict(loss_cls=torch.tensor([1.0, 3.0]), acc=torch.tensor(0.5)))
    assert float(loss) == 2.0 and logs["loss"] == 2.0 and logs["acc"] == 0.5


def test_checkpoint_roundtrip_module_prefix_and_nonstrict(tmp_path):
    import mvfnet_amd
    from mvfnet_amd.checkpoint import load_checkpoint, save_checkpoint
    m = mvfnet_amd.build_recognizer(mvfnet_amd.mvfnet_config(50, 4))
    path = save_checkpoint(m, str(tmp_path / "epoch_1.pth"), optimizer=dict(x=1), meta=dict(epoch=1, iter=7))
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"meta", "state_dict", "optimizer"} and ck["meta"]["epoch"] == 1
    assert "backbone.layer3.0.conv1.shift_conv.weight" in ck["state_dict"]        # the released-checkpoint key layout
    # a DataParallel-style checkpoint ('module.' prefix) with one tensor missing and one extra loads non-strictly
    sd = {"module." + k: v + 1.0 if v.dtype.is_floating_point else v for k, v in ck["state_dict"].items()}
    sd.pop("module.cls_head.new_fc.bias")
    sd["module.extra.weight"] = torch.zeros(1)
    torch.save(dict(state_dict=sd), str(tmp_path / "dp.pth"))
    m2 = mvfnet_amd.build_recognizer(mvfnet_amd.mvfnet_config(50, 4))
    before = m2.cls_head.new_fc.bias.clone()
    msgs = []
    class L:
        def warning(self, t):
            msgs.append(t)
    load_checkpoint(m2, str(tmp_path / "dp.pth"), logger=L())
    assert torch.equal(m2.backbone.conv1.weight, m.backbone.conv1.weight + 1.0)
    assert torch.equal(m2.cls_head.new_fc.bias, before)
    assert "extra.weight" in msgs[0] and "cls_head.new_fc.bias" in msgs[0]
    with pytest.raises(RuntimeError):
        load_checkpoint(m2, str(tmp_path / "dp.pth"), strict=True)
    with pytest.raises(IOError):
        load_checkpoint(m2, "https://example.com/x.pth")
    # pretrained= path in the backbone config goes through the same loader (ImageNet weights land in conv1.weight BEFORE
    # the MVF wrapper exists and are re-homed as conv1.net.weight, reference recognizer2d.py:45-59)
    plain = {k[len("backbone."):].replace(".conv1.net.", ".conv1."): v for k, v in ck["state_dict"].items()
             if k.startswith("backbone.") and "shift_conv" not in k and "h_conv" not in k and "w_conv" not in k and ".conv1.bn." not in k}
    torch.save(plain, str(tmp_path / "resnet50.pth"))
    cfg = mvfnet_amd.mvfnet_config(50, 4)
    cfg["backbone"]["pretrained"] = str(tmp_path / "resnet50.pth")
    m3 = mvfnet_amd.build_recognizer(cfg)
    assert torch.equal(m3.backbone.layer3[0].conv1.net.weight, m.backbone.layer3[0].conv1.net.weight)


# ------------------------------------------------------------------------------------------------ optimizer wire format
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _our_block():
    import torch.nn as nn
    from mvfnet_amd.backbones.resnet import Bottleneck
    from mvfnet_amd.modules import MVF
    blk = Bottleneck(64, 16, 1, 1, None)
    blk.conv1 = MVF(blk.conv1, 4, 64, 0.125, True, False, "THW")
    return blk


@pytest.mark.parametrize("tag", ["plain", "paramwise"])
def test_reference_written_checkpoint_loads_and_its_optimizer_state_maps(tag):
    """tests/golden/ref_ckpt_block_*.pth were WRITTEN BY THE REFERENCE (its save_checkpoint + its build_optimizer's torch SGD,
    tests/golden/make_ckpt_golden.py): weights load key for key, the optimizer entry maps to per-parameter momentum buffers in
    this repo's parameter order, and what this repo writes back is loadable by torch.optim.SGD with the same group structure."""
    import numpy as np
    from mvfnet_amd.checkpoint import load_checkpoint, sgd_momentum_buffers, sgd_state_dict
    from mvfnet_amd.runner import paramwise_multipliers
    blk = _our_block()
    ck = load_checkpoint(blk, os.path.join(GOLDEN, "ref_ckpt_block_%s.pth" % tag), strict=True)
    assert set(ck) == {"meta", "state_dict", "optimizer"} and ck["meta"]["epoch"] == 3 and ck["meta"]["iter"] == 14
    g = np.load(os.path.join(GOLDEN, "ref_ckpt_block.npz"))
    for k, v in blk.state_dict().items():
        assert np.array_equal(v.numpy(), g["%s/%s" % (tag, k)]), k
    params = list(blk.parameters())
    bufs, group = sgd_momentum_buffers(ck["optimizer"], len(params))
    assert len(bufs) == len(params) and all(b is not None and tuple(b.shape) == tuple(p.shape) for b, p in zip(bufs, params))
    assert group["momentum"] == 0.9 and group["nesterov"] is True
    mult = None
    if tag == "paramwise":
        m = paramwise_multipliers(blk, dict(bias_lr_mult=2.0, bias_decay_mult=0.0, norm_decay_mult=0.0))
        mult = [m.get(p, (1.0, 1.0)) for p in params]
        # the reference's groups carry the multiplied values: same classification of every parameter
        for grp, (a, b) in zip(ck["optimizer"]["param_groups"], mult):
            assert abs(grp["lr"] - 0.015 * a) < 1e-12 and abs(grp["weight_decay"] - 1e-4 * b) < 1e-12
    ours = sgd_state_dict(bufs, 0.015, 0.9, 1e-4, True, multipliers=mult)
    assert len(ours["param_groups"]) == len(ck["optimizer"]["param_groups"])
    # torch's own SGD, built the way the reference builds it, accepts what this repo writes
    if mult is None:
        opt = torch.optim.SGD(blk.parameters(), lr=0.015, momentum=0.9, weight_decay=1e-4, nesterov=True)
    else:
        opt = torch.optim.SGD([dict(params=[p], lr=0.015 * a, weight_decay=1e-4 * b) for p, (a, b) in zip(params, mult)], lr=0.015,
                              momentum=0.9, weight_decay=1e-4, nesterov=True)
    opt.load_state_dict(ours)
    for p, b in zip(params, bufs):
        assert torch.equal(opt.state[p]["momentum_buffer"], b)
    ref_sd = opt.state_dict()
    assert [grp["params"] for grp in ref_sd["param_groups"]] == [grp["params"] for grp in ours["param_groups"]]


def test_model_parameter_order_matches_the_reference():
    """torch optimizer state is keyed by position in model.parameters(): the order must be the reference's."""
    import numpy as np
    import mvfnet_amd
    g = np.load(os.path.join(GOLDEN, "ref_ckpt_block.npz"))
    for depth, t in ((50, 8), (101, 16)):
        m = mvfnet_amd.build_recognizer(mvfnet_amd.mvfnet_config(depth, t))
        assert [k for k, _ in m.named_parameters()] == [str(x) for x in g["r%d/param_order" % depth]]


def test_build_optimizer_argument_checks_need_no_gpu():
    from mvfnet_amd.runner import Config, build_optimizer, paramwise_multipliers
    blk = _our_block()
    with pytest.raises(NotImplementedError):
        build_optimizer(blk, dict(type="Adam", lr=1e-3))
    with pytest.raises(ValueError):                 # decay multiplier without an explicit weight_decay (reference train.py:123-125)
        build_optimizer(blk, dict(type="SGD", lr=0.1, paramwise_options=dict(norm_decay_mult=0.0)))
    m = paramwise_multipliers(blk, dict(bias_lr_mult=2.0, bias_decay_mult=0.5, norm_decay_mult=0.0))
    names = {id(p): k for k, p in blk.named_parameters()}
    got = {names[id(p)]: v for p, v in m.items()}
    assert got["bn1.weight"] == (1.0, 0.0) and got["bn3.bias"] == (1.0, 0.0) and "conv2.weight" not in got
    assert got["conv1.bn.weight"] == (1.0, 0.0)            # the MVF's BatchNorm3d is named `bn`: matches (bn|gn)(\d+)?.(weight|bias)
    c = Config(optimizer=dict(type="SGD", lr=0.015), data=dict(videos_per_gpu=4))
    assert c.optimizer.lr == 0.015 and c.data.videos_per_gpu == 4 and c.get("fp16") is None


def test_device_prefetcher_passes_host_batches_through_without_a_gpu():
    from mvfnet_amd.runner import DevicePrefetcher
    batches = [dict(img_group=torch.full((2, 3), float(i)), label=torch.tensor([i, i])) for i in range(3)]
    out = list(DevicePrefetcher(batches, device="cpu"))
    assert len(out) == 3 and all(torch.equal(a["img_group"], b["img_group"]) for a, b in zip(out, batches))
