"""W4A4 GEMM wrapper (reference: nunchaku/ops/gemm.py:12-160).  Keyword names are the reference's."""

from __future__ import annotations

import math

import torch

from .._C import ops


def svdq_gemm_w4a4_cuda(
    act: torch.Tensor,
    wgt: torch.Tensor,
    out: torch.Tensor | None = None,
    qout: torch.Tensor | None = None,
    ascales: torch.Tensor | None = None,
    wscales: torch.Tensor | None = None,
    oscales: torch.Tensor | None = None,
    poolout: torch.Tensor | None = None,
    lora_act_in: torch.Tensor | None = None,
    lora_up: torch.Tensor | None = None,
    lora_down: torch.Tensor | None = None,
    lora_act_out: torch.Tensor | None = None,
    norm_q: torch.Tensor | None = None,
    norm_k: torch.Tensor | None = None,
    rotary_emb: torch.Tensor | None = None,
    bias: torch.Tensor | None = None,
    smooth_factor: torch.Tensor | None = None,
    out_vk: torch.Tensor | None = None,
    out_linearattn: torch.Tensor | None = None,
    act_unsigned: bool = False,
    lora_scales: list[float] | None = None,
    fuse_silu: bool = False,
    fp4: bool = False,
    alpha: float | None = 1.0,
    wcscales: torch.Tensor | None = None,
    out_q: torch.Tensor | None = None,
    out_k: torch.Tensor | None = None,
    out_v: torch.Tensor | None = None,
    attn_tokens: int = 0,
    out_vt: torch.Tensor | None = None,
    lora_act_zeroed: bool = False,
    second: dict | None = None,
    split_rows: int = 0,
) -> None:
    """Fused W4A4 GEMM + low-rank correction; results are written in place into ``out`` or, for the
    GELU+requantise fusion, into ``qout`` / ``oscales`` / ``lora_act_out``."""
    if lora_scales is None:
        rank = 0 if lora_up is None else lora_up.shape[1]
        lora_scales = [1.0] * math.ceil(rank / 16)
    if alpha is None:
        alpha = 1.0
    ops.gemm_w4a4(
        act, wgt, out, qout, ascales, wscales, oscales, poolout, lora_act_in, lora_up, lora_down, lora_act_out,
        norm_q, norm_k, rotary_emb, bias, smooth_factor, out_vk, out_linearattn, act_unsigned, lora_scales,
        fuse_silu, fp4, alpha, wcscales, out_q, out_k, out_v, attn_tokens, out_vt, lora_act_zeroed, second, split_rows,
    )
