"""Data parallelism for the sparse-conv hot path: scenes shard one-per-GPU, gradients are summed by one
flat-bucket all-reduce (RCCL over xGMI on the GPU, gloo in the CPU tests).

The reference has no collective code of its own; scenes never interact (the batch index is
part of the hash key), so the only exchange step of a training iteration is the sum of the weight / bias
gradients.  Those are small (0.88 MB per 64->128 layer), i.e. latency / per-link bound on point-to-point
xGMI: everything is flattened into as few buckets as possible so a step issues one collective, not one per
parameter.  Kernel selection is a pure function of shapes (no run-time autotune), so ranks cannot diverge.
"""
from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def rank_and_world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_scenes(num_scenes: int, rank: Optional[int] = None, world_size: Optional[int] = None) -> List[int]:
    """Scene i goes to rank i mod W.  Returns the scene ids owned by ``rank``."""
    if rank is None or world_size is None:
        rank, world_size = rank_and_world()
    return list(range(rank, num_scenes, world_size))


@torch.no_grad()
def allreduce_gradients(params: Iterable[torch.nn.Parameter], group=None, average: bool = True,
                        bucket_bytes: int = 256 << 20) -> int:
    """Sum (or average) ``p.grad`` over the ranks with flat buckets; returns the number of collectives issued.

    Parameters without a gradient contribute zeros so every rank issues identical collectives.
    """
    params = [p for p in params if p.requires_grad]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1 or not params:
        return 0
    world = dist.get_world_size(group)
    calls = 0
    by_dtype = {}
    for p in params:
        by_dtype.setdefault((p.dtype if p.grad is None else p.grad.dtype, p.device), []).append(p)
    for (dtype, device), plist in by_dtype.items():
        bucket: List[torch.nn.Parameter] = []
        size = 0

        def flush(bucket):
            flat = torch.cat([(q.grad if q.grad is not None else torch.zeros_like(q, dtype=dtype)).reshape(-1) for q in bucket])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            if average:
                flat.div_(world)
            off = 0
            for q in bucket:
                n = q.numel()
                g = flat[off : off + n].view_as(q)
                if q.grad is None:
                    q.grad = g.clone()
                else:
                    q.grad.copy_(g)
                off += n

        for p in plist:
            nbytes = p.numel() * p.element_size()
            if bucket and size + nbytes > bucket_bytes:
                flush(bucket)
                calls += 1
                bucket, size = [], 0
            bucket.append(p)
            size += nbytes
        if bucket:
            flush(bucket)
            calls += 1
    return calls


def claim_grad_slot(param) -> Optional[torch.Tensor]:
    """The bucket slot of ``param`` for ONE producer per backward pass, else None.

    A weight-gradient kernel may write into the slot instead of a fresh tensor only if it is the first producer for this
    parameter in this backward pass: with the parameter used by several nodes of one graph (weight tying, a block applied
    twice, siamese branches) ``param.grad`` is still None when every node runs - AccumulateGrad fires after ALL producers -
    so a second writer would overwrite the first and the engine would sum two aliases of the same memory (N x the last
    gradient instead of the sum).  The claim is released by the post-accumulate hook, ``finish()`` and ``zero_grad()``; it also
    carries the id of the autograd graph task that made it, so a claim left behind by a backward pass that never reached
    AccumulateGrad (``torch.autograd.grad``, an exception in the middle of a backward) is recognised as stale by the next pass
    instead of silently switching the write-into-bucket path off for good."""
    slot = getattr(param, "_wcn_grad_slot", None)
    if slot is None or param.grad is not None:
        return None
    task = _graph_task_id()
    held = getattr(param, "_wcn_grad_claimed", False)
    if held is not False and (task < 0 or held == ("task", task)):
        return None  # claimed in THIS backward pass (or no way to tell the passes apart)
    param._wcn_grad_claimed = ("task", task)
    return slot


def _graph_task_id() -> int:
    """Id of the autograd graph task being executed (-1 outside a backward pass / on builds without the accessor)."""
    fn = getattr(torch._C, "_current_graph_task_id", None)
    try:
        return int(fn()) if fn is not None else -1
    except Exception:  # pragma: no cover
        return -1


class GradientBuckets:
    """Persistent flat gradient buckets with the all-reduce overlapped with the rest of the backward pass.

    * every parameter's ``.grad`` is a VIEW into one flat buffer per bucket - no ``cat`` before the collective, no copy
      back after it (a gradient that arrives as a fresh tensor, e.g. after ``zero_grad(set_to_none=True)``, is moved into
      its view by the hook);
    * the view is also published as ``param._wcn_grad_slot``: a producer that can write its result anywhere - the sparse
      convolution's weight-gradient kernels - writes STRAIGHT into the bucket when ``param.grad is None`` and hands autograd an
      alias of the slot, which the engine adopts as ``.grad`` (no ``grad += dw`` launch per parameter, no copy by the hook).
      Only the FIRST producer of a parameter per backward pass gets the slot (``claim_grad_slot``); a parameter used twice
      in one graph has its further gradients written to fresh tensors and summed by the engine as usual.
      ``zero_grad()`` therefore sets the gradients to None by default (``set_to_none=False`` zero-fills in place);
    * buckets are filled in REVERSE parameter order (the order the backward pass produces gradients) and a bucket's
      all-reduce is launched from the autograd hook of its last gradient, asynchronously, so RCCL works over xGMI while
      the remaining layers are still back-propagating;
    * launches are strictly in bucket order and ``finish()`` launches whatever the hooks did not (parameters that took
      no part in this iteration contribute zeros), so every rank issues the same collectives in the same order.

    Bucket size: xGMI is point-to-point (7 links x ~153 GB/s per GPU), a ring all-reduce is per-link bound and the
    parameter sets of this path are small (0.88 MB per 64->128 layer, 10-40 MB per MinkUNet) - the default 32 MB makes a
    whole network one or two collectives.

        buckets = GradientBuckets(model.parameters())
        loss.backward()          # collectives start inside
        buckets.finish()         # wait, average; p.grad is ready for the optimizer
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None, average: bool = True, bucket_bytes: int = 32 << 20,
                 collective_when_alone: bool = False):
        # collective_when_alone: issue the all-reduce even in a group of ONE rank (a no-op sum that still goes through the
        # backend's streams, events and work handles) - how the single-GPU test box exercises the RCCL path of this class
        self.group, self.average = group, average
        self._alone_too = bool(collective_when_alone) and dist.is_available() and dist.is_initialized()
        self.params = [p for p in params if p.requires_grad]
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self._buckets: List[dict] = []
        self._where: Dict[int, tuple] = {}
        order = list(reversed(self.params))
        by_kind: Dict[tuple, List[torch.nn.Parameter]] = {}
        for p in order:
            by_kind.setdefault((p.dtype, p.device), []).append(p)
        for (dtype, device), plist in by_kind.items():
            cur, size = [], 0
            for p in plist:
                nbytes = p.numel() * p.element_size()
                if cur and size + nbytes > bucket_bytes:
                    self._make_bucket(cur, dtype, device)
                    cur, size = [], 0
                cur.append(p)
                size += nbytes
            if cur:
                self._make_bucket(cur, dtype, device)
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]
        self._accumulate = False
        self._reset()

    def _make_bucket(self, plist, dtype, device):
        total = sum(p.numel() for p in plist)
        flat = torch.zeros(total, dtype=dtype, device=device)
        b = {"flat": flat, "params": plist, "views": [], "ready": 0, "launched": False, "work": None}
        off = 0
        for p in plist:
            v = flat[off : off + p.numel()].view_as(p)
            b["views"].append(v)
            self._where[id(p)] = (len(self._buckets), len(b["views"]) - 1)
            p.grad = v
            p._wcn_grad_slot = v
            off += p.numel()
        self._buckets.append(b)

    def _reset(self):
        for b in self._buckets:
            b["ready"], b["launched"], b["work"], b["seen"] = 0, False, None, set()
        self._next = 0

    def _launch_ready(self):
        # strictly in bucket order: identical collective sequence on every rank
        while self._next < len(self._buckets):
            b = self._buckets[self._next]
            if b["ready"] < len(b["params"]):
                return
            self._launch(b)
            self._next += 1

    def _launch(self, b):
        b["launched"] = True
        if self.world > 1 or self._alone_too:
            b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    @torch.no_grad()
    def _hook(self, p):
        p._wcn_grad_claimed = False  # every producer of this backward pass has run
        bi, vi = self._where[id(p)]
        b = self._buckets[bi]
        v = b["views"][vi]
        if p.grad is not v:
            if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)  # the gradient arrived as a fresh tensor: move it into the bucket
            p.grad = v
        if self._accumulate:
            return  # no_sync(): gradients pile up in the buckets, nothing is launched until the next synchronised backward
        if b["launched"] or id(p) in b["seen"]:
            # a second backward() before finish(): its gradients were added into a buffer whose all-reduce is already in
            # flight (or done) and would never be reduced - refuse instead of training on silently wrong gradients
            raise RuntimeError(
                "GradientBuckets: a parameter received a second gradient before finish() - call finish() after every "
                "synchronised backward(), and wrap the accumulation micro-steps in `with buckets.no_sync():`")
        b["seen"].add(id(p))
        b["ready"] += 1
        self._launch_ready()

    def no_sync(self):
        """Context manager for gradient accumulation: backward passes inside it only accumulate into the buckets (no
        collective is launched); the first backward outside it - followed by ``finish()`` - reduces the sum."""
        outer = self

        class _NoSync:
            def __enter__(self_inner):
                self_inner.prev = outer._accumulate
                outer._accumulate = True

            def __exit__(self_inner, *exc):
                outer._accumulate = self_inner.prev
                return False

        return _NoSync()

    def remove(self):
        """Detach the autograd hooks (the gradients stay views of the buckets until they are reassigned)."""
        for h in self._handles:
            h.remove()
        self._handles = []
        for p in self.params:
            if hasattr(p, "_wcn_grad_slot"):
                del p._wcn_grad_slot
            p._wcn_grad_claimed = False

    @torch.no_grad()
    def finish(self) -> int:
        """Launch what is left (zeros for parameters without a gradient this iteration), wait, average.  Returns the
        number of collectives of this iteration."""
        for p in self.params:
            p._wcn_grad_claimed = False
        for b in self._buckets:
            if not b["launched"]:
                for p, v in zip(b["params"], b["views"]):
                    if id(p) not in b["seen"]:
                        if p.grad is not None and p.grad is not v and p.grad.data_ptr() != v.data_ptr():
                            v.copy_(p.grad)
                        elif p.grad is None:
                            v.zero_()
                        p.grad = v
                self._launch(b)
        self._next = len(self._buckets)
        calls = 0
        for b in self._buckets:
            if b["work"] is not None:
                b["work"].wait()
                calls += 1
                if self.average:
                    b["flat"].div_(self.world)
        self._reset()
        return calls

    def zero_grad(self, set_to_none: bool = True):
        """``set_to_none`` (default, what ``optimizer.zero_grad()`` does too): gradients become None - the next backward writes
        the weight gradients of the sparse convolutions straight into their bucket slots, every other gradient is moved in by
        the hook, parameters without a gradient contribute zeros (``finish``).  False: zero the buckets in place and keep the
        views attached (gradient accumulation across ``no_sync()`` micro-steps needs neither)."""
        for p in self.params:
            p._wcn_grad_claimed = False
        if set_to_none:
            for p in self.params:
                p.grad = None
            return
        for b in self._buckets:
            b["flat"].zero_()
        for p in self.params:
            bi, vi = self._where[id(p)]
            p.grad = self._buckets[bi]["views"][vi]
