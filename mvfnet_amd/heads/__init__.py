from .tsn_clshead import TSNClsHead

__all__ = ["TSNClsHead"]
