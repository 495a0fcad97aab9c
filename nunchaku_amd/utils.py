"""Small helpers shared by the op wrappers (reference: nunchaku/utils.py:17-58,113-131,216-222)."""

from __future__ import annotations

import torch


def ceil_divide(x: int, divisor: int) -> int:
    return (x + divisor - 1) // divisor


def pad_tensor(tensor: torch.Tensor | None, multiples: int, dim: int, fill=0) -> torch.Tensor | None:
    """Pad ``tensor`` along ``dim`` up to the next multiple of ``multiples`` (None passes through)."""
    if tensor is None or multiples <= 1:
        return tensor
    size = tensor.shape[dim]
    target = ceil_divide(size, multiples) * multiples
    if target == size:
        return tensor
    shape = list(tensor.shape)
    shape[dim] = target
    out = tensor.new_full(shape, fill)
    out.narrow(dim, 0, size).copy_(tensor)
    return out


def get_precision(precision: str = "auto", device="cuda", pretrained_model_name_or_path=None) -> str:
    """gfx950 has no NVFP4 path: always "int4" (the reference returns "fp4" only on SM 120/121)."""
    if precision not in ("auto", "int4", "fp4"):
        raise ValueError(f"invalid precision {precision!r}")
    if precision == "fp4":
        raise NotImplementedError("NVFP4 checkpoints are Blackwell-only; load the int4 checkpoint on MI355X")
    return "int4"
