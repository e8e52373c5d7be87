"""Feature batch base (reference `geometry/base/features.py:10-32`)."""
from .batched import BatchedTensor


class Features(BatchedTensor):
    @property
    def num_channels(self) -> int:
        return self.batched_tensor.shape[-1]

    @property
    def is_cat(self) -> bool:
        return True

    @property
    def is_pad(self) -> bool:
        return False
