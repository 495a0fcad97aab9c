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

deterministic = False


def set_deterministic(flag: bool = True) -> None:
    global deterministic
    deterministic = bool(flag)


@contextlib.contextmanager
def deterministic_mode(flag: bool = True):
    global deterministic
    old, deterministic = deterministic, bool(flag)
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
