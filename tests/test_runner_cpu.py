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
