from .cat import CatFeatures, to_batched_features

__all__ = ["CatFeatures", "to_batched_features"]
