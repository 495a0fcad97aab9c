"""W4A4 GEMM wrapper (reference: nunchaku/ops/gemm.py:12-160).  Keyword names and order are the reference's;
everything after ``attn_tokens`` is an extension of this library."""

from __future__ import annotations

import math

import torch

from .._C import ops

T = torch.Tensor


def svdq_gemm_w4a4_cuda(
    act: T, wgt: T, out: T | None = None, qout: T | None = None,            # operand images, 16-bit / requantised output
    ascales: T | None = None, wscales: T | None = None, oscales: T | None = None, poolout: T | None = None,
    lora_act_in: T | None = None, lora_up: T | None = None,                 # low-rank branch of THIS layer
    lora_down: T | None = None, lora_act_out: T | None = None,              # ... of the NEXT layer (GELU_QUANT epilogue)
    norm_q: T | None = None, norm_k: T | None = None, rotary_emb: T | None = None,   # RMSNorm + RoPE epilogue
    bias: T | None = None, smooth_factor: T | None = None,
    out_vk: T | None = None, out_linearattn: T | None = None,               # SANA LiteLA: not supported
    act_unsigned: bool = False, lora_scales: list[float] | None = None, fuse_silu: bool = False,
    fp4: bool = False, alpha: float | None = 1.0, wcscales: T | None = None,          # NVFP4 only
    out_q: T | None = None, out_k: T | None = None, out_v: T | None = None, attn_tokens: int = 0,
    out_vt: T | None = None, lora_act_zeroed: bool = False, second: dict | None = None, split_rows: int = 0,
    q_scale: float = 0.0,
) -> None:
    """Fused W4A4 GEMM + low-rank correction; results are written in place into ``out`` or, for the
    GELU+requantise fusion, into ``qout`` / ``oscales`` / ``lora_act_out``.  Extensions: ``out_vt`` (transposed V side
    output of the RoPE epilogue), ``lora_act_zeroed`` (skip the clear of ``lora_act_out``), ``second`` / ``split_rows``
    (a second weight set for the rows from ``split_rows`` on: two streams in one launch), ``q_scale`` (RoPE epilogue: the Q third
    times this factor before its rounding to 16-bit -- for ``ops.attention(q_prescaled=True)``)."""
    if lora_scales is None:  # one scale per 16 ranks, all ones (reference :125-127)
        lora_scales = [1.0] * math.ceil((0 if lora_up is None else lora_up.shape[1]) / 16)
    ops.gemm_w4a4(
        act, wgt, out, qout, ascales, wscales, oscales, poolout, lora_act_in, lora_up, lora_down, lora_act_out,
        norm_q, norm_k, rotary_emb, bias, smooth_factor, out_vk, out_linearattn, act_unsigned, lora_scales,
        fuse_silu, fp4, 1.0 if alpha is None else alpha, wcscales, out_q, out_k, out_v, attn_tokens,
        out_vt, lora_act_zeroed, second, split_rows, q_scale,
    )
