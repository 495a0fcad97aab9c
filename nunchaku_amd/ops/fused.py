"""Fused layer helpers (reference: nunchaku/ops/fused.py:14-79, 82-178)."""

from __future__ import annotations

import torch

from ..utils import ceil_divide
from .gemm import svdq_gemm_w4a4_cuda


def fused_gelu_mlp(x: torch.Tensor, fc1, fc2, pad_size: int = 256, ln=None) -> torch.Tensor:
    """MLP ``fc2(gelu(fc1(x)))`` in three launches: quantise, fc1 GEMM whose epilogue applies GELU,
    re-quantises to unsigned 4-bit (shift 0.171875) and computes fc2's low-rank down projection,
    then the fc2 GEMM on those codes."""
    B, S, C_in = x.shape
    M = B * S
    x2 = x.reshape(M, C_in)
    qx, ascales, lora_act = fc1.quantize(x2, ln=ln)  # ln: fused AdaLayerNormZero front end (extension)
    M_pad = ceil_divide(M, pad_size) * pad_size
    dev = x.device
    q_hidden = torch.empty(M_pad, fc1.out_features * 3 // 4, dtype=torch.uint8, device=dev)  # FP6 operand image
    s_hidden = torch.empty(fc1.out_features // 64, M_pad, dtype=x.dtype, device=dev)
    pool = ln[3] if ln is not None and len(ln) > 3 else None  # scratch cleared by the preceding element-wise pass
    l_hidden = pool.take(M_pad * fc2.proj_down.shape[1]) if pool is not None else None
    l_zeroed = l_hidden is not None
    l_hidden = l_hidden.view(M_pad, -1) if l_zeroed else torch.empty(M_pad, fc2.proj_down.shape[1], dtype=torch.float32, device=dev)
    fc1._ensure_layout()
    fc2._ensure_layout()
    svdq_gemm_w4a4_cuda(
        act=qx, wgt=fc1.qweight, qout=q_hidden, ascales=ascales, wscales=fc1.wscales, oscales=s_hidden,
        lora_act_in=lora_act, lora_up=fc1.proj_up, lora_down=fc2.proj_down, lora_act_out=l_hidden,
        bias=fc1.bias, smooth_factor=fc2.smooth_factor, fp4=False, alpha=fc1.wtscale, wcscales=fc1.wcscales,
        lora_scales=getattr(fc1, "lora_scales", None), lora_act_zeroed=l_zeroed,
    )
    out = torch.empty(M, fc2.out_features, dtype=x.dtype, device=dev)
    out = fc2.forward_quant(q_hidden, s_hidden, l_hidden, output=out)
    return out.view(B, S, -1)


def fused_qkv_norm_rottary(x: torch.Tensor, proj, norm_q=None, norm_k=None, rotary_emb: torch.Tensor | None = None,
                           output=None, attn_tokens: int = 0, out_vt: torch.Tensor | None = None, ln=None):
    """QKV projection with RMSNorm(q), RMSNorm(k) and rotary embedding applied in the GEMM epilogue.
    ``rotary_emb`` is the ``pack_rotemb`` tensor of the reference ([1, M_pad, 128] float32).
    ``out_vt`` ([out_features/3, tokens] view): V is written transposed there for ``ops.attention`` instead of
    into ``output`` (this library's form of the reference's ``output=(q, k, v)`` packed mode)."""
    B, S, C_in = x.shape
    M = B * S
    x2 = x.reshape(M, C_in)
    qx, ascales, lora_act = proj.quantize(x2, ln=ln)
    if isinstance(output, tuple):
        raise NotImplementedError("the reference's packed (q, k, v) tuple is NVIDIA-fragment ordered; pass out_vt= instead")
    if output is None:
        output = torch.empty(M, proj.out_features, dtype=x.dtype, device=x.device)
    proj._ensure_layout()
    rot = None
    if rotary_emb is not None:
        rot = rotary_emb.reshape(-1, rotary_emb.shape[-1])
    svdq_gemm_w4a4_cuda(
        act=qx, wgt=proj.qweight, out=output, ascales=ascales, wscales=proj.wscales, lora_act_in=lora_act,
        lora_up=proj.proj_up, bias=proj.bias, fp4=False, alpha=proj.wtscale, wcscales=proj.wcscales,
        norm_q=None if norm_q is None else norm_q.weight, norm_k=None if norm_k is None else norm_k.weight,
        rotary_emb=rot, out_vt=out_vt, lora_scales=getattr(proj, "lora_scales", None),
    )
    return output.view(B, S, -1)
