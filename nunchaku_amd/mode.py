"""Process-wide numerics mode of the op wrappers.

``deterministic``: the low-rank activations ``lora_act`` are accumulated as Q31.32 fixed point with 64-bit integer atomics
(``SVDQ_LORA_ACT_Q32``, include/svdq_amd.h "lora_act formats") instead of fp32 atomics: integer addition is associative,
so K-sliced quantiser sums, the GELU epilogue's column-tile sums and the attention epilogue's head sums no longer depend
on arrival order and a model forward is bit-reproducible from run to run.  The reference is NOT deterministic here (fp32
atomics, lora.cuh:82-94,323), and neither is the default mode of this package (the fast path: half the bytes per value).
The format travels with the tensor: a ``torch.int64`` ``lora_act`` IS the fixed-point format, ``torch.float32`` the fp32 one
-- producers and consumers (``_C.ops``) read it off the dtype, so buffers allocated in one mode stay valid after a switch.
"""

from __future__ import annotations

import contextlib

import torch

deterministic = False  # False | True ("strict") | "runs"


def _level(flag):
    """``False`` | ``True`` / ``"strict"`` | ``"runs"``.

    strict: the fixed-point sums do not depend on the launch configuration either (tile geometry, grid, stream-K) -- every partial sum is an integer atomic.
    runs (``SVDQ_LORA_ACT_Q32_RUNS``, ABI 22): a GELU_QUANT launch sums the column tiles one workgroup walks in a row in fp32, in a fixed order, and adds the
    run's sum as fixed point: bit-reproducible from run to run and between replicas for a given shape and device -- what the mode is used for -- at about half
    the step-time cost of strict; not equal to the sums of another launch geometry."""
    if flag in (False, None, 0):
        return False
    if flag == "runs":
        return "runs"
    if flag in (True, 1, "strict"):
        return True
    raise ValueError(f"deterministic level {flag!r}: False, True / 'strict' or 'runs'")


def set_deterministic(flag=True) -> None:
    global deterministic
    deterministic = _level(flag)


def gemm_lora_act_format_runs() -> bool:
    """the GEMM wrapper passes SVDQ_LORA_ACT_Q32_RUNS for int64 low-rank buffers"""
    return deterministic == "runs"


@contextlib.contextmanager
def deterministic_mode(flag=True):
    global deterministic
    old, deterministic = deterministic, _level(flag)
    try:
        yield
    finally:
        deterministic = old


def lora_act_words() -> int:
    """fp32 words per lora_act element in the current mode"""
    return 2 if deterministic else 1


def alloc_lora_act(rows: int, R: int, device, pool=None):
    """``(lora_act [rows, R], already_zeroed)`` in the current mode's format; ``pool``: a ZeroPool of pre-cleared fp32 words."""
    words = lora_act_words()
    piece = pool.take(rows * R * words) if pool is not None else None
    if piece is not None:
        return (piece.view(torch.int64) if deterministic else piece).view(rows, R), True
    return torch.empty(rows, R, dtype=torch.int64 if deterministic else torch.float32, device=device), False


def lora_act_to_float(t: torch.Tensor) -> torch.Tensor:
    """fp32 view / conversion of a lora_act tensor of either format (tests, debugging)."""
    return t if t.dtype == torch.float32 else (t.double() * 2.0 ** -32).float()
