from .resnet import Bottleneck, ResNet, make_res_layer

__all__ = ["ResNet", "Bottleneck", "make_res_layer"]
