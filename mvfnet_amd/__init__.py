"""mvfnet_amd -- MI355X-native MVFNet hot path (MVF module + ResNet conv stack + TSN head) behind the reference's
module / recognizer API.  Importing the model classes registers them (Recognizer2D, ResNet, TSNClsHead)."""

__version__ = "0.1.0"


def _register():
    from . import backbones, heads, recognizers  # noqa: F401


def build_recognizer(cfg, train_cfg=None, test_cfg=None):
    _register()
    from .builder import build_recognizer as _b
    return _b(cfg, train_cfg, test_cfg)


def mvfnet_config(depth=50, n_segment=8, num_classes=400, dropout_ratio=0.5, fcn_testing=False, alpha=0.125,
                  mvf_freq=(0, 0, 1, 1), mode="THW"):
    """The `model = dict(...)` of configs/MVFNet/K400/mvf_kinetics400_2d_rgb_r{50,101}_dense.py:20-48."""
    return dict(
        type="Recognizer2D",
        backbone=dict(type="ResNet", pretrained=None, depth=depth, out_indices=(3,), norm_eval=False, partial_norm=False,
                      norm_cfg=dict(type="BN", requires_grad=True)),
        cls_head=dict(type="TSNClsHead", spatial_size=-1, spatial_type="avg", with_avg_pool=False, temporal_feature_size=1,
                      spatial_feature_size=1, dropout_ratio=dropout_ratio, in_channels=2048, init_std=0.01,
                      num_classes=num_classes, fcn_testing=fcn_testing),
        fcn_testing=fcn_testing,
        module_cfg=dict(type="MVF", n_segment=n_segment, alpha=alpha, mvf_freq=mvf_freq, mode=mode))
