"""build_recognizer / build_backbone / build_head (reference: codes/models/builder.py:6-37)."""
import torch.nn as nn

from .registry import Registry, build_from_cfg

RECOGNIZERS = Registry("recognizer")
BACKBONES = Registry("backbone")
HEADS = Registry("head")


def build(cfg, registry, default_args=None):
    if isinstance(cfg, list):
        return nn.Sequential(*[build_from_cfg(c, registry, default_args) for c in cfg])
    return build_from_cfg(cfg, registry, default_args)


def build_recognizer(cfg, train_cfg=None, test_cfg=None):
    return build(cfg, RECOGNIZERS, dict(train_cfg=train_cfg, test_cfg=test_cfg))


def build_backbone(cfg):
    return build(cfg, BACKBONES)


def build_head(cfg):
    return build(cfg, HEADS)
