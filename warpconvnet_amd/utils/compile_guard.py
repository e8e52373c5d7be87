"""``@eager_unless_compiling``: `torch.compiler.disable` semantics without its eager-mode cost.

The orchestration entry points are data dependent (host offsets, kernel-map cache, ctypes calls) and must stay out of
`torch.compile` graphs (reference: `@torch.compiler.disable` on `helper.py:147, 361`, `torch_discrete.py:294`).  The
stock decorator routes EVERY call through dynamo's eval-frame wrapper - 20-30 us per call in eager mode, ~70 calls per
MinkUNet iteration.  This wrapper calls the function directly in eager mode and takes the disabled path only while a
compiler is tracing (`torch.compiler.is_compiling()` is a constant for the tracer, so the graph breaks exactly as before).
"""
import functools

import torch


def eager_unless_compiling(fn):
    disabled = torch.compiler.disable(fn)

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        if torch.compiler.is_compiling():
            return disabled(*args, **kwargs)
        return fn(*args, **kwargs)

    return wrapper
