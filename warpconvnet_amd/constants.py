"""Environment-variable flag surface for the sparse-conv hot path.

Mirrors the three algorithm knobs of the reference (`warpconvnet/constants.py:116-162`):
``WARPCONVNET_{FWD,DGRAD,WGRAD}_ALGO_MODE``.  Valid values here are the build's own backends:

* ``auto``          - static choice: the fused HIP kernels (no run-time sweep, so ranks never diverge)
* ``hip_mfma``      - fused gather->MFMA->store kernels (bf16/fp16/fp32), gfx950
* ``hip_ref``       - simple one-thread-per-output HIP kernels (any channel count; slow, for bring-up)
* ``explicit_gemm`` - gather / torch.matmul / index_add on the tensor's device (reference `explicit.py` semantics)
"""
import os
from typing import List, Optional

VALID_ALGOS = ["auto", "hip_mfma", "hip_ref", "explicit_gemm"]


def _env_choice(name: str, default: str, valid: Optional[List[str]] = None) -> str:
    v = os.environ.get(name)
    if v is None:
        return default
    v = v.strip().lower()
    if valid is not None and v not in valid:
        raise ValueError(f"{name} must be one of {valid}, got {v!r}")
    return v


def _env_bool(name: str, default: bool) -> bool:
    v = os.environ.get(name)
    if v is None:
        return default
    v = v.strip().lower()
    if v not in ("true", "false", "1", "0"):
        raise ValueError(f"{name} must be one of true/false/1/0, got {v!r}")
    return v in ("true", "1")


WARPCONVNET_FWD_ALGO_MODE = _env_choice("WARPCONVNET_FWD_ALGO_MODE", "auto", VALID_ALGOS)
WARPCONVNET_DGRAD_ALGO_MODE = _env_choice("WARPCONVNET_DGRAD_ALGO_MODE", "auto", VALID_ALGOS)
WARPCONVNET_WGRAD_ALGO_MODE = _env_choice("WARPCONVNET_WGRAD_ALGO_MODE", "auto", VALID_ALGOS)

# MFMA on CDNA4 always accumulates in fp32; the flag exists for API parity
# (reference `constants.py:214, 251-271`) and is ignored by the kernels.
_USE_FP16_ACCUM = _env_bool("WARPCONVNET_USE_FP16_ACCUM", False)


def get_fp16_accum() -> bool:
    return _USE_FP16_ACCUM


def set_fp16_accum(enabled: bool) -> None:
    global _USE_FP16_ACCUM
    _USE_FP16_ACCUM = bool(enabled)

# Depthwise convolution backends (reference `constants.py:159-172`; generic names are accepted as aliases)
VALID_DEPTHWISE_ALGOS = ["explicit", "implicit", "explicit_gemm", "implicit_gemm", "auto"]
WARPCONVNET_DEPTHWISE_CONV_FWD_ALGO_MODE = _env_choice("WARPCONVNET_DEPTHWISE_CONV_FWD_ALGO_MODE", "auto", VALID_DEPTHWISE_ALGOS)
WARPCONVNET_DEPTHWISE_CONV_BWD_ALGO_MODE = _env_choice("WARPCONVNET_DEPTHWISE_CONV_BWD_ALGO_MODE", "auto", VALID_DEPTHWISE_ALGOS)
