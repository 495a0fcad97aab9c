"""Element-wise glue of a transformer block with the next LayerNorm's statistics fused in (extension of the
reference surface; the reference's V2 blocks use torch ops here: transformer_flux_v2.py:118-342)."""

from __future__ import annotations

import torch

from .._C import ops


def residual_gate_stats(res: torch.Tensor, a: torch.Tensor | None = None, gate: torch.Tensor | None = None,
                        b: torch.Tensor | None = None, inplace: bool = True, want_stats: bool = True, eps: float = 1e-6):
    """``y = res + gate * (a [+ b])`` (one 16-bit rounding per torch op, as the reference's blocks; ``a`` None: ``y = res``) and the row
    statistics ``[rows, 2]`` float32 (mean, rstd) of ``y`` for a following ``quantize(..., ln=...)``.
    Tensors are ``[..., C]`` contiguous; returns ``(y, stats)``."""
    C = res.shape[-1]
    r2 = res.reshape(-1, C)
    out = None
    if a is not None:
        out = r2 if inplace else torch.empty_like(r2)
    stats = torch.empty(r2.shape[0], 2, dtype=torch.float32, device=res.device) if want_stats else None
    ops.residual_gate_stats(r2, None if a is None else a.reshape(-1, C), None if b is None else b.reshape(-1, C),
                            None if gate is None else gate.reshape(-1), out, stats, eps)
    y = res if out is None else out.view(res.shape)
    return y, stats
