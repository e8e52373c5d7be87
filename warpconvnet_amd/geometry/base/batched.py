"""Concatenated ragged batch: one [N, C] tensor + CPU ``offsets[B+1]``.

API follows the reference `warpconvnet/geometry/base/batched.py:14-270`: list or (tensor, offsets)
constructor, ``offsets`` always on the host, ``to()/half()/float()`` return new objects, integer
indexing returns one batch element's rows.
"""
from typing import List, Optional, Sequence, Union

import torch
from torch import Tensor

from warpconvnet_amd.geometry.utils.list_to_batch import list_to_cat_tensor


def _host_offsets(offsets) -> Tensor:
    if isinstance(offsets, Tensor):
        off = offsets.detach().cpu()
        if off.dtype not in (torch.int32, torch.int64):
            off = off.to(torch.int64)
        return off
    if isinstance(offsets, Sequence):
        return torch.tensor(list(offsets), dtype=torch.int64)
    raise ValueError(f"Invalid offsets type {type(offsets)}")


class BatchedTensor:
    batched_tensor: Tensor
    offsets: Tensor

    def __init__(
        self,
        batched_tensor: Union[List[Tensor], Tensor],
        offsets: Optional[Union[List[int], Tensor]] = None,
        device: Optional[Union[str, torch.device]] = None,
    ):
        if isinstance(batched_tensor, (list, tuple)):
            assert offsets is None, "If batched_tensors is a list, offsets must be None."
            batched_tensor, offsets, _ = list_to_cat_tensor(batched_tensor)
        else:
            assert isinstance(batched_tensor, Tensor), "Batched tensor must be a tensor or a list"
            if offsets is None:
                offsets = [0, batched_tensor.shape[0]]
        self.offsets = _host_offsets(offsets)
        if device is not None:
            batched_tensor = batched_tensor.to(device)
        self.batched_tensor = batched_tensor
        self.check()

    # -- invariants -------------------------------------------------------
    def check(self):
        assert self.offsets.device.type == "cpu" and self.offsets.dtype in (
            torch.int32,
            torch.int64,
        ), f"Offsets must be a cpu int tensor, got {self.offsets}"
        assert not self.offsets.requires_grad
        assert isinstance(self.batched_tensor, Tensor)

    # -- basic properties -------------------------------------------------
    @property
    def batch_size(self) -> int:
        return len(self.offsets) - 1

    @property
    def device(self):
        return self.batched_tensor.device

    @property
    def shape(self):
        return self.batched_tensor.shape

    @property
    def dtype(self):
        return self.batched_tensor.dtype

    def numel(self):
        return self.batched_tensor.numel()

    def __len__(self) -> int:
        return self.batched_tensor.shape[0]

    def __getitem__(self, idx: int) -> Tensor:
        if isinstance(idx, int):
            if idx < 0:
                idx += self.batch_size
            return self.batched_tensor[int(self.offsets[idx]) : int(self.offsets[idx + 1])]
        raise TypeError(f"unsupported index {idx!r}")

    # -- conversions (always new objects) ----------------------------------
    def _new(self, tensor: Tensor) -> "BatchedTensor":
        out = object.__new__(self.__class__)
        out.__dict__.update(self.__dict__)
        out.__dict__.pop("_bcoords", None)
        out.batched_tensor = tensor
        return out

    def to(self, device=None, dtype: Optional[torch.dtype] = None) -> "BatchedTensor":
        if isinstance(device, torch.dtype):  # allow .to(torch.float16)
            device, dtype = None, device
        return self._new(self.batched_tensor.to(device=device or self.device, dtype=dtype))

    def half(self):
        return self._new(self.batched_tensor.half())

    def float(self):
        return self._new(self.batched_tensor.float())

    def double(self):
        return self._new(self.batched_tensor.double())

    def clone(self):
        return self._new(self.batched_tensor.clone())

    def to_nested(self):
        return torch.nested.as_nested_tensor([self[i] for i in range(self.batch_size)])

    def equal_shape(self, value: "BatchedTensor") -> bool:
        return bool((self.offsets == value.offsets).all()) and self.numel() == value.numel()

    def equal_rigorous(self, value: "BatchedTensor") -> bool:
        return self.equal_shape(value) and bool((self.batched_tensor == value.batched_tensor).all())

    # -- arithmetic with scalars / one-element tensors / batches of the same shape (reference batched.py:190-238) --
    def binary_op(self, value: object, op: str) -> "BatchedTensor":
        if isinstance(value, (int, float)) or (torch.is_tensor(value) and value.numel() == 1):
            return self._new(getattr(self.batched_tensor, op)(value))
        assert isinstance(value, BatchedTensor) and self.equal_shape(value), "operands must be batches of the same shape"
        return self._new(getattr(self.batched_tensor, op)(value.batched_tensor))

    def __add__(self, value):
        return self.binary_op(value, "__add__")

    def __sub__(self, value):
        return self.binary_op(value, "__sub__")

    def __mul__(self, value):
        return self.binary_op(value, "__mul__")

    def __truediv__(self, value):
        return self.binary_op(value, "__truediv__")

    def __floordiv__(self, value):
        return self.binary_op(value, "__floordiv__")

    def __mod__(self, value):
        return self.binary_op(value, "__mod__")

    def __pow__(self, value):
        return self.binary_op(value, "__pow__")

    def __repr__(self) -> str:
        return (
            f"{self.__class__.__name__}(offsets={self.offsets.tolist()}, "
            f"shape={tuple(self.batched_tensor.shape)}, device={self.device}, dtype={self.dtype})"
        )
