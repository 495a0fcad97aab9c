"""Element-wise glue of a transformer block with the next LayerNorm's statistics fused in (extension of the
reference surface; the reference's V2 blocks use torch ops here: transformer_flux_v2.py:118-342)."""

from __future__ import annotations

import torch

from .._C import ops
from ..mode import lora_act_words


class ZeroPool:
    """Scratch words (fp32-sized) cleared by a :func:`residual_gate_stats` pass, handed out in pieces to the quantiser /
    GELU_QUANT calls that follow it on the same stream (their low-rank accumulators need a zeroed buffer: this saves
    one memset launch each).  ``take`` returns None when the pool is exhausted -- the caller then clears its own."""

    def __init__(self, buf: torch.Tensor):
        self.buf, self.used = buf, 0

    def take(self, numel: int):
        n = (numel + 3) // 4 * 4  # keep every piece 16-byte aligned
        if self.used + n > self.buf.numel():
            return None
        piece = self.buf[self.used:self.used + numel]
        self.used += n
        return piece


def residual_gate_stats(res: torch.Tensor, a: torch.Tensor | None = None, gate: torch.Tensor | None = None,
                        b: torch.Tensor | None = None, inplace: bool = True, want_stats: bool = True, eps: float = 1e-6,
                        zero_floats: int = 0, clamp_fp16: bool = False):
    """``y = res + gate * (a [+ b])`` (one 16-bit rounding per torch op, as the reference's blocks; ``a`` None: ``y = res``) and the row
    statistics ``[rows, 2]`` float32 (mean, rstd) of ``y`` for a following ``quantize(..., ln=...)``.  ``clamp_fp16``: clip
    ``y`` to +-65504 when the dtype is fp16 (the reference's fp16 blocks do, transformer_flux_v2.py:254-255, 339-340).
    Tensors are ``[..., C]`` contiguous; returns ``(y, stats)`` or, with ``zero_floats`` > 0, ``(y, stats, ZeroPool)``
    where the pool holds that many fp32 zeros cleared in the same pass."""
    C = res.shape[-1]
    r2 = res.reshape(-1, C)
    out = None
    if a is not None:
        out = r2 if inplace else torch.empty_like(r2)
    stats = torch.empty(r2.shape[0], 2, dtype=torch.float32, device=res.device) if want_stats else None
    want_pool = zero_floats > 0
    zero_floats *= lora_act_words()  # the count is in lora_act elements: two words each in deterministic mode
    zero = torch.empty((zero_floats + 3) // 4 * 4, dtype=torch.float32, device=res.device) if zero_floats > 0 else None
    ops.residual_gate_stats(r2, None if a is None else a.reshape(-1, C), None if b is None else b.reshape(-1, C),
                            None if gate is None else gate.reshape(-1), out, stats, eps, zero, clamp_fp16=int(bool(clamp_fp16)))
    y = res if out is None else out.view(res.shape)
    if want_pool:
        return y, stats, ZeroPool(zero if zero is not None else torch.empty(0, dtype=torch.float32, device=res.device))
    return y, stats


def residual_gate_stats_pair(res_a, a_a, gate_a, res_b, a_b, gate_b, zero_floats: int = 0, eps: float = 1e-6, clamp_fp16_a: bool = False,
                             clamp_fp16_b: bool = False):
    """Two independent gated residuals (the two streams of a joint block: same width, different row counts) and their
    statistics in ONE launch, both in place.  Returns ``(y_a, stats_a, y_b, stats_b[, ZeroPool])``."""
    C = res_a.shape[-1]
    ra, rb = res_a.reshape(-1, C), res_b.reshape(-1, C)
    sa = torch.empty(ra.shape[0], 2, dtype=torch.float32, device=res_a.device)
    sb = torch.empty(rb.shape[0], 2, dtype=torch.float32, device=res_a.device)
    zf = zero_floats * lora_act_words()
    zero = torch.empty((zf + 3) // 4 * 4, dtype=torch.float32, device=res_a.device) if zf > 0 else None
    ops.residual_gate_stats(ra, a_a.reshape(-1, C), None, gate_a.reshape(-1), ra, sa, eps, zero,
                            second=(rb, a_b.reshape(-1, C), None, gate_b.reshape(-1), rb, sb),
                            clamp_fp16=int(bool(clamp_fp16_a)) | (2 if clamp_fp16_b else 0))
    if zero_floats > 0:
        return res_a, sa, res_b, sb, ZeroPool(zero if zero is not None else torch.empty(0, dtype=torch.float32, device=res_a.device))
    return res_a, sa, res_b, sb
