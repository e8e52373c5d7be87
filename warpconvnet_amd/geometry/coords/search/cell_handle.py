"""Device tables cached on coordinate tensors, guarded against in-place edits.

The block-hashed cell table of a coordinate set (`csrc/kmap_cells.h`) and the kernel map a down-sampling pass wrote next
to its output coordinates are kept as attributes of the coordinate TENSOR so that later layers on the same tensor reuse
them.  A tensor edited in place afterwards (augmentation shift, ``coords.add_``) must not be answered from a table of its
old content: every handle records ``tensor._version`` and is ignored once the counter has moved.  The reference has no
such cache - its maps are keyed per call (`warpconvnet/nn/functional/sparse_conv/helper.py:446-459`) - so a stale hit would
be a behavioural difference, not an optimisation.
"""
import weakref
from typing import Optional, Tuple

from torch import Tensor


def _version(t: Tensor) -> Optional[int]:
    try:
        return t._version
    except RuntimeError:  # inference tensors carry no version counter: never cached
        return None


def attach_cells(t: Tensor, ws: Tensor, n: int, max_blocks: int) -> None:
    v = _version(t)
    if v is not None:
        t._wcn_cells = (ws, int(n), int(max_blocks), v)


def cells_of(t: Tensor) -> Optional[Tuple[Tensor, int, int]]:
    """(workspace, n, max_blocks) of the validated cell table of `t`'s CURRENT content, or None."""
    h = getattr(t, "_wcn_cells", None)
    if h is None:
        return None
    if h[1] != t.shape[0] or h[3] != _version(t):
        try:
            del t._wcn_cells  # stale: edited in place (or resized) since the build
        except AttributeError:
            pass
        return None
    return h[0], h[1], h[2]


def attach_stride_map(out: Tensor, src: Tensor, stride, nbr: Tensor, mask: Tensor) -> None:
    vo, vs = _version(out), _version(src)
    if vo is not None and vs is not None:
        out._wcn_stride_map = (weakref.ref(src), vs, vo, int(src.shape[0]), tuple(int(v) for v in stride), nbr, mask)


def stride_map_of(out: Tensor, *srcs: Tensor):
    """(stride, nbr, mask) of the map the down-sampling of one of `srcs` left on `out`, if both are unchanged since."""
    h = getattr(out, "_wcn_stride_map", None)
    if h is None:
        return None
    src = h[0]()
    if (src is None or not any(src is s for s in srcs) or h[1] != _version(src) or h[2] != _version(out)
            or h[3] != src.shape[0]):
        return None
    return h[4], h[5], h[6]
