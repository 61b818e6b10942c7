from .MVF import MVF, HardSwish, make_multi_view_fusion

__all__ = ["MVF", "HardSwish", "make_multi_view_fusion"]
