"""torch.compile glue (reference `warpconvnet/_compile.py:29-143`, `helper.py:147, 361`, `torch_discrete.py:294`).

``torch.compile(model)`` must keep working for models that contain sparse convolutions: the orchestration functions are
data dependent (host offsets, kernel-map cache, ctypes calls into libwcn_hip.so), so they are excluded from tracing
(`torch.compiler.disable`: the compiled graph breaks around them), the autograd Functions are allowed in the graph, and
``IntSearchResult`` is a pytree node (children: in_maps, out_maps, offsets).
"""
import torch
from torch.utils._pytree import register_pytree_node

from warpconvnet_amd.geometry.coords.search.search_results import IntSearchResult


def _flatten(obj: IntSearchResult):
    return [obj.in_maps, obj.out_maps, obj.offsets], {"identity_map_index": obj.identity_map_index}


def _unflatten(children, ctx) -> IntSearchResult:
    # no constructor call: its offsets[-1] check would read a tensor value under tracing
    self = object.__new__(IntSearchResult)
    self._in_maps, self._out_maps, self._offsets = children
    self._lazy_pairs = None
    self._num_offsets = len(children[2]) - 1
    self._init_tables()  # device tables are rebuilt from the CSR form on first use
    self.identity_map_index = ctx["identity_map_index"]
    return self


_DONE = False


def register() -> None:
    global _DONE
    if _DONE:
        return
    _DONE = True
    register_pytree_node(IntSearchResult, _flatten, _unflatten)
    import torch._dynamo

    from warpconvnet_amd.nn.functional.sparse_conv.detail.unified import UnifiedSpatiallySparseConvFunction
    from warpconvnet_amd.nn.functional.sparse_conv_depth import UnifiedSpatiallySparseDepthwiseConvFunction
    from warpconvnet_amd.nn.functional.sparse_pool import _SparsePoolFunction
    from warpconvnet_amd.ops.reductions import _SegmentReduce

    for fn in (UnifiedSpatiallySparseConvFunction, UnifiedSpatiallySparseDepthwiseConvFunction, _SparsePoolFunction,
               _SegmentReduce):
        torch._dynamo.allow_in_graph(fn)
