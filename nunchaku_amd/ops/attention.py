"""Attention over the packed QKV buffer (role of the reference's ``nunchaku-fp16`` attention processor:
nunchaku/models/attention_processors/flux.py:62-137 -> ops.attention_fp16, src/kernels/zgemm/attention.cu:11-94)."""

from __future__ import annotations

import math

import torch

from .._C import ops


def alloc_qkv(tokens: int, heads: int, dtype: torch.dtype, device, head_dim: int = 128):
    """Buffers one attention layer needs: ``qkv`` [tokens, 3*H*D] (the V third stays unused) and ``vt`` [H*D, tokens]."""
    qkv = torch.empty(tokens, 3 * heads * head_dim, dtype=dtype, device=device)
    vt = torch.empty(heads * head_dim, tokens, dtype=dtype, device=device)
    return qkv, vt


def attention_packed(qkv: torch.Tensor, vt: torch.Tensor, heads: int, out: torch.Tensor | None = None,
                     scale: float | None = None, zero_floats: int = 0):
    """``softmax(scale * Q K^T) V`` per head, reading Q and K in place from the fused QKV GEMM output
    ``qkv`` [L, 3*H*128] and V from its transposed side output ``vt`` [H*128, L]; returns ``[L, H*128]``
    token-major (the layout the output projection's quantiser reads).  No transposes, no copies.
    ``zero_floats`` > 0: also returns a ``ZeroPool`` of that many fp32 zeros cleared by the same launch (the low-rank
    accumulators of the output projections' quantisers): ``(out, pool)``."""
    L, three_hd = qkv.shape
    D = three_hd // (3 * heads)
    if three_hd != 3 * heads * D or tuple(vt.shape) != (heads * D, L):
        raise ValueError("attention_packed: expected qkv [L, 3*H*D] and vt [H*D, L]")
    if out is None:
        out = torch.empty(L, heads * D, dtype=qkv.dtype, device=qkv.device)
    q = qkv[:, : heads * D].unflatten(1, (heads, D))
    k = qkv[:, heads * D : 2 * heads * D].unflatten(1, (heads, D))
    zero = torch.empty((zero_floats + 3) // 4 * 4, dtype=torch.float32, device=qkv.device) if zero_floats > 0 else None
    ops.attention(q, k, vt.unflatten(0, (heads, D)), out.unflatten(1, (heads, D)), 1.0 / math.sqrt(D) if scale is None else scale, zero)
    if zero_floats > 0:
        from .elementwise import ZeroPool

        return out, ZeroPool(zero)
    return out
