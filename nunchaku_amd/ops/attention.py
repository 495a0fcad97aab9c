"""Attention over the packed QKV buffer (role of the reference's ``nunchaku-fp16`` attention processor:
nunchaku/models/attention_processors/flux.py:62-137 -> ops.attention_fp16, src/kernels/zgemm/attention.cu:11-94)."""

from __future__ import annotations

import math

import torch

from .._C import ops
from ..mode import alloc_lora_act, lora_act_words


def alloc_qkv(tokens: int, heads: int, dtype: torch.dtype, device, head_dim: int = 128):
    """Buffers one attention layer needs: ``qkv`` [tokens, 3*H*D] (the V third stays unused) and ``vt`` [H*D, tokens]."""
    qkv = torch.empty(tokens, 3 * heads * head_dim, dtype=dtype, device=device)
    vt = torch.empty(heads * head_dim, tokens, dtype=dtype, device=device)
    return qkv, vt


def q_prescale(head_dim: int = 128, scale: float | None = None) -> float:
    """The factor a QKV GEMM applies to Q (``fused_qkv_norm_rottary(..., q_scale=)``) so that the attention kernel needs no per-score
    scaling: ``scale * log2(e)``.  Pass ``q_prescaled=True`` to the attention call that reads that buffer."""
    return (1.0 / math.sqrt(head_dim) if scale is None else scale) * 1.4426950408889634


def kv_valid_ranges(t_first: int, t_second: int = 0, pad: int = 256):
    """Key ranges of a [first | second] token sequence whose streams are each padded to ``pad`` rows (what the grouped launches need: every
    stream starts on a 256-row boundary): ``None`` when nothing is padded, ``(n,)`` when only the tail is, ``(n0, start1, end1)`` when the
    padding of the first stream sits in the middle -- the forms ``ops.attention(kv_valid=...)`` takes."""
    p_first = (t_first + pad - 1) // pad * pad
    p_second = (t_second + pad - 1) // pad * pad
    if p_first == t_first and p_second == t_second:
        return None
    if t_second == 0:
        return (t_first,)
    if p_first == t_first:
        return (t_first + t_second,)
    return (t_first, p_first, p_first + t_second)


def attention_packed(qkv: torch.Tensor, vt: torch.Tensor, heads: int, out: torch.Tensor | None = None,
                     scale: float | None = None, zero_floats: int = 0, q_prescaled: bool = False, kv_valid=None):
    """``softmax(scale * Q K^T) V`` per head, reading Q and K in place from the fused QKV GEMM output
    ``qkv`` [L, 3*H*128] and V from its transposed side output ``vt`` [H*128, L]; returns ``[L, H*128]``
    token-major (the layout the output projection's quantiser reads).  No transposes, no copies.
    ``zero_floats`` > 0: also returns a ``ZeroPool`` of that many fp32 zeros cleared by the same launch (the low-rank
    accumulators of the output projections' quantisers): ``(out, pool)``.  ``kv_valid``: the real key rows of a padded buffer
    (:func:`kv_valid_ranges`); the other keys get probability 0."""
    L, three_hd = qkv.shape
    D = three_hd // (3 * heads)
    if three_hd != 3 * heads * D or tuple(vt.shape) != (heads * D, L):
        raise ValueError("attention_packed: expected qkv [L, 3*H*D] and vt [H*D, L]")
    if out is None:
        out = torch.empty(L, heads * D, dtype=qkv.dtype, device=qkv.device)
    q = qkv[:, : heads * D].unflatten(1, (heads, D))
    k = qkv[:, heads * D : 2 * heads * D].unflatten(1, (heads, D))
    zwords = zero_floats * lora_act_words()
    zero = torch.empty((zwords + 3) // 4 * 4, dtype=torch.float32, device=qkv.device) if zero_floats > 0 else None
    ops.attention(q, k, vt.unflatten(0, (heads, D)), out.unflatten(1, (heads, D)), 1.0 / math.sqrt(D) if scale is None else scale, zero,
                  q_prescaled=q_prescaled, kv_valid=kv_valid)
    if zero_floats > 0:
        from .elementwise import ZeroPool

        return out, ZeroPool(zero)
    return out


def attention_packed_quantized(qkv: torch.Tensor, vt: torch.Tensor, heads: int, lin, lin_first=None, split_rows: int = 0,
                               pool=None, scale: float | None = None, q_prescaled: bool = False, kv_valid=None):
    """Attention whose epilogue emits the quantised input of the output projection ``lin`` directly (codes, scales and
    low-rank down projection: what ``lin.quantize(attention_packed(...))`` would return, without the 16-bit round trip).
    Joint attention: rows ``< split_rows`` belong to ``lin_first`` (text), the rest to ``lin``.  ``pool``: a ZeroPool for
    the low-rank accumulator.  Returns ``(codes, scales, lora_act)`` or None when the shapes do not allow it."""
    L, three_hd = qkv.shape
    D = three_hd // (3 * heads)
    K, R = heads * D, lin.rank
    if D != 128 or L % 256 or K != lin.in_features or R > 256 or R % 16 or (lin_first is not None and (
            lin_first.rank != R or lin_first.in_features != K or split_rows % 256 or not 0 < split_rows < L)):
        return None
    lin._ensure_layout()
    dev = qkv.device
    act = torch.empty(L, K * 3 // 4, dtype=torch.uint8, device=dev)
    asc = torch.empty(K // 64, L, dtype=qkv.dtype, device=dev)
    lact, zeroed = alloc_lora_act(L, R, dev, pool)
    if not zeroed:
        lact.zero_()
    quant = dict(act=act, ascales=asc, lora_act=lact, R=R)
    if lin_first is not None:  # first parameter set = the rows that come first (text)
        lin_first._ensure_layout()
        quant.update(smooth=lin_first.smooth_factor, lora_down=lin_first.proj_down, smooth2=lin.smooth_factor,
                     lora_down2=lin.proj_down, split_rows=split_rows)
    else:
        quant.update(smooth=lin.smooth_factor, lora_down=lin.proj_down)
    q = qkv[:, : heads * D].unflatten(1, (heads, D))
    k = qkv[:, heads * D : 2 * heads * D].unflatten(1, (heads, D))
    ops.attention(q, k, vt.unflatten(0, (heads, D)), None, 1.0 / math.sqrt(D) if scale is None else scale, None, quant, q_prescaled=q_prescaled,
                  kv_valid=kv_valid)
    return act, asc, lact
