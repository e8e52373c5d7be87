"""k-nearest-neighbour search: chunked cdist + topk per batch element.

Same algorithm as the reference (`warpconvnet/geometry/coords/search/knn.py:11-26, 108-142`, pure
torch there too); indices are global row ids (local index + batch offset).
"""
import torch
from torch import Tensor


@torch.no_grad()
def knn_search(ref: Tensor, query: Tensor, k: int, chunk: int = 4096) -> Tensor:
    """[M, k] int64 indices into ``ref`` of the k nearest reference points of every query."""
    assert k <= ref.shape[0], f"knn_k={k} exceeds the number of reference points {ref.shape[0]}"
    out = []
    for s in range(0, query.shape[0], chunk):
        d = torch.cdist(query[s : s + chunk], ref)
        out.append(torch.topk(d, k, dim=1, largest=False).indices)
    return torch.cat(out, 0) if out else torch.zeros((0, k), dtype=torch.int64, device=ref.device)


@torch.no_grad()
def batched_knn_search(ref: Tensor, ref_offsets: Tensor, query: Tensor, query_offsets: Tensor, k: int,
                       chunk: int = 4096) -> Tensor:
    assert len(ref_offsets) == len(query_offsets)
    outs = []
    for b in range(len(ref_offsets) - 1):
        r0, r1 = int(ref_offsets[b]), int(ref_offsets[b + 1])
        q0, q1 = int(query_offsets[b]), int(query_offsets[b + 1])
        outs.append(knn_search(ref[r0:r1], query[q0:q1], k, chunk) + r0)
    return torch.cat(outs, 0)
