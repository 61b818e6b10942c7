from .recognizer2d import Recognizer2D

__all__ = ["Recognizer2D"]
