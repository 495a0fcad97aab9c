"""Fused layer helpers (reference: nunchaku/ops/fused.py:14-79, 82-178)."""

from __future__ import annotations

import torch

from ..mode import alloc_lora_act
from ..utils import ceil_divide
from .gemm import svdq_gemm_w4a4_cuda


def fused_gelu_mlp(x: torch.Tensor, fc1, fc2, pad_size: int = 256, ln=None, quantized=None) -> torch.Tensor:
    """MLP ``fc2(gelu(fc1(x)))`` in three launches: quantise, fc1 GEMM whose epilogue applies GELU,
    re-quantises to unsigned 4-bit (shift 0.171875) and computes fc2's low-rank down projection,
    then the fc2 GEMM on those codes."""
    B, S, C_in = x.shape
    M = B * S
    x2 = x.reshape(M, C_in)
    # ln: fused AdaLayerNormZero front end (extension); quantized: (codes, scales, lora_act) made by quantize_two
    qx, ascales, lora_act = fc1.quantize(x2, ln=ln) if quantized is None else quantized
    M_pad = ceil_divide(M, pad_size) * pad_size
    dev = x.device
    q_hidden = torch.empty(M_pad, fc1.out_features * 3 // 4, dtype=torch.uint8, device=dev)  # FP6 operand image
    s_hidden = torch.empty(fc1.out_features // 64, M_pad, dtype=x.dtype, device=dev)
    pool = ln[3] if ln is not None and len(ln) > 3 else None  # scratch cleared by the preceding element-wise pass
    l_hidden, l_zeroed = alloc_lora_act(M_pad, fc2.proj_down.shape[1], dev, pool)
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
                           output=None, attn_tokens: int = 0, out_vt: torch.Tensor | None = None, ln=None, quantized=None,
                           q_scale: float = 0.0):
    """QKV projection with RMSNorm(q), RMSNorm(k) and rotary embedding applied in the GEMM epilogue.
    ``rotary_emb`` is the ``pack_rotemb`` tensor of the reference ([1, M_pad, 128] float32).
    ``out_vt`` ([out_features/3, tokens] view): V is written transposed there for ``ops.attention`` instead of
    into ``output`` (this library's form of the reference's ``output=(q, k, v)`` packed mode).  ``q_scale``: Q is emitted times this
    factor (``ops.attention.q_prescale()``) for ``attention_packed(..., q_prescaled=True)``."""
    B, S, C_in = x.shape
    M = B * S
    x2 = x.reshape(M, C_in)
    qx, ascales, lora_act = proj.quantize(x2, ln=ln) if quantized is None else quantized
    if isinstance(output, tuple):
        # the reference's "nunchaku-fp16" attention hand-off (ops/fused.py:140-160): three opaque [B, H, T_pad, 128] buffers
        # that only _C.ops.attention_fp16 reads.  Adapter path (one scatter copy); out_vt= is the copy-free form.
        assert len(output) == 3
        proj._ensure_layout()
        svdq_gemm_w4a4_cuda(
            act=qx, wgt=proj.qweight, ascales=ascales, wscales=proj.wscales, lora_act_in=lora_act, lora_up=proj.proj_up,
            bias=proj.bias, fp4=False, alpha=proj.wtscale, wcscales=proj.wcscales,
            norm_q=None if norm_q is None else norm_q.weight, norm_k=None if norm_k is None else norm_k.weight,
            rotary_emb=None if rotary_emb is None else rotary_emb.reshape(-1, rotary_emb.shape[-1]),
            out_q=output[0], out_k=output[1], out_v=output[2], attn_tokens=attn_tokens or M,
            lora_scales=getattr(proj, "lora_scales", None))
        return output
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
        rotary_emb=rot, out_vt=out_vt, lora_scales=getattr(proj, "lora_scales", None), q_scale=q_scale,
    )
    return output.view(B, S, -1)


# --------------------------------------------------------------------------------------------------------------------
# Grouped launches for the two streams of a joint block (extension): stream a (text) owns rows [0, Ma), stream b
# (image) the rows after it; same layer shapes, different weights.  The row-side buffers of the two streams are
# allocated back to back, each stream is quantised into its slice, and ONE gemm_w4a4 launch with a second weight set
# serves both (include/svdq_amd.h: svdq_gemm_args.wgt2 ...).  Ma must be a multiple of 256 and batch 1.
# --------------------------------------------------------------------------------------------------------------------
from .quantize import svdq_quantize_w4a4_act_fuse_lora_cuda  # noqa: E402


def _pair_ok(la, lb, xa, xb) -> bool:
    return (xa.shape[0] == 1 and xb.shape[0] == 1 and xa.shape[1] % 256 == 0 and la.in_features == lb.in_features
            and la.out_features == lb.out_features and la.rank == lb.rank and (la.bias is None) == (lb.bias is None)
            and la.act_unsigned == lb.act_unsigned and getattr(la, "lora_scales", None) == getattr(lb, "lora_scales", None))


def _quantize_pair(xa, la, xb, lb, ln_a=None, ln_b=None, pool=None):
    """Both streams quantised into one set of row-side buffers (stream a first).  Returns (act, ascales, lora_act, Ma)."""
    la._ensure_layout()
    lb._ensure_layout()
    Ma, Mb, K, R = xa.shape[1], xb.shape[1], la.in_features, la.rank
    Mb_pad = ceil_divide(Mb, 256) * 256
    Mt, dev = Ma + Mb_pad, xa.device
    act = torch.empty(Mt, K * 3 // 4, dtype=torch.uint8, device=dev)
    asc = torch.empty(K // 64, Mt, dtype=xa.dtype, device=dev)  # opaque scale image: row-tile major, so the streams are slices
    lact, zeroed = alloc_lora_act(Mt, R, dev, pool)
    # ONE quantiser launch: stream a's rows first, stream b read from its own tensor with its own parameter set
    sec = dict(input=xb, smooth=lb.smooth_factor, lora_down=lb.proj_down)
    if ln_b is not None:
        sec.update(ln_stats=ln_b[0], mod_scale=ln_b[1], mod_shift=ln_b[2])
    svdq_quantize_w4a4_act_fuse_lora_cuda(
        xa.reshape(-1, K), output=act, oscales=asc, lora_down=la.proj_down, lora_act_out=lact, smooth=la.smooth_factor,
        ln=None if ln_a is None else ln_a[:3], lora_act_zeroed=zeroed, second=sec)
    return act, asc, lact, Ma


def _second(lin, **extra):
    d = dict(wgt=lin.qweight, wscales=lin.wscales, bias=lin.bias, lora_up=lin.proj_up)
    d.update(extra)
    return d


def linear_pair(xa, la, xb, lb, pool=None):
    """``(la(xa), lb(xb))`` with one GEMM launch; falls back to two calls when the layers cannot be grouped."""
    if not _pair_ok(la, lb, xa, xb):
        return la(xa, pool=pool), lb(xb, pool=pool)
    act, asc, lact, Ma = _quantize_pair(xa, la, xb, lb, pool=pool)
    Mb = xb.shape[1]
    out = torch.empty(Ma + Mb, la.out_features, dtype=xa.dtype, device=xa.device)
    svdq_gemm_w4a4_cuda(act=act, wgt=la.qweight, out=out, ascales=asc, wscales=la.wscales, lora_act_in=lact, lora_up=la.proj_up,
                        bias=la.bias, act_unsigned=la.act_unsigned, lora_scales=getattr(la, "lora_scales", None),
                        second=_second(lb), split_rows=Ma)
    return out[:Ma].unsqueeze(0), out[Ma:].unsqueeze(0)


def linear_pair_quantized(act, asc, lact, la, lb, Ma):
    """The GEMM half of :func:`linear_pair` on an already quantised joint activation (rows < Ma: layer la)."""
    la._ensure_layout()
    lb._ensure_layout()
    Mt = act.shape[0]
    out = torch.empty(Mt, la.out_features, dtype=asc.dtype, device=act.device)
    svdq_gemm_w4a4_cuda(act=act, wgt=la.qweight, out=out, ascales=asc, wscales=la.wscales, lora_act_in=lact, lora_up=la.proj_up,
                        bias=la.bias, act_unsigned=la.act_unsigned, lora_scales=getattr(la, "lora_scales", None),
                        second=_second(lb), split_rows=Ma)
    return out[:Ma].unsqueeze(0), out[Ma:].unsqueeze(0)


def fused_gelu_mlp_pair(xa, fc1a, fc2a, xb, fc1b, fc2b, ln_a=None, ln_b=None):
    """``(fc2a(gelu(fc1a(xa))), fc2b(gelu(fc1b(xb))))`` with two GEMM launches instead of four."""
    if not (_pair_ok(fc1a, fc1b, xa, xb) and _pair_ok(fc2a, fc2b, xa, xb)):
        return fused_gelu_mlp(xa, fc1a, fc2a, ln=ln_a), fused_gelu_mlp(xb, fc1b, fc2b, ln=ln_b)
    pool = ln_a[3] if ln_a is not None and len(ln_a) > 3 else None  # stream a's pool must hold both streams' scratch
    act, asc, lact, Ma = _quantize_pair(xa, fc1a, xb, fc1b, ln_a, ln_b, pool=pool)
    for m in (fc2a, fc2b):
        m._ensure_layout()
    Mb, Mt, dev = xb.shape[1], act.shape[0], xa.device
    Nh, R2 = fc1a.out_features, fc2a.rank
    q_hidden = torch.empty(Mt, Nh * 3 // 4, dtype=torch.uint8, device=dev)
    s_hidden = torch.empty(Nh // 64, Mt, dtype=xa.dtype, device=dev)
    l_hidden, l_zeroed = alloc_lora_act(Mt, R2, dev, pool)
    svdq_gemm_w4a4_cuda(
        act=act, wgt=fc1a.qweight, qout=q_hidden, ascales=asc, wscales=fc1a.wscales, oscales=s_hidden, lora_act_in=lact,
        lora_up=fc1a.proj_up, lora_down=fc2a.proj_down, lora_act_out=l_hidden, bias=fc1a.bias, smooth_factor=fc2a.smooth_factor,
        lora_scales=getattr(fc1a, "lora_scales", None), lora_act_zeroed=l_zeroed,
        second=_second(fc1b, smooth_factor=fc2b.smooth_factor, lora_down=fc2b.proj_down), split_rows=Ma)
    out = torch.empty(Ma + Mb, fc2a.out_features, dtype=xa.dtype, device=dev)
    svdq_gemm_w4a4_cuda(act=q_hidden, wgt=fc2a.qweight, out=out, ascales=s_hidden, wscales=fc2a.wscales, lora_act_in=l_hidden,
                        lora_up=fc2a.proj_up, bias=fc2a.bias, act_unsigned=fc2a.act_unsigned,
                        lora_scales=getattr(fc2a, "lora_scales", None), second=_second(fc2b), split_rows=Ma)
    return out[:Ma].unsqueeze(0), out[Ma:].unsqueeze(0)


def fused_qkv_norm_rottary_pair(xa, proj_a, nq_a, nk_a, xb, proj_b, nq_b, nk_b, rotary_emb, output, out_vt=None,
                                ln_a=None, ln_b=None, q_scale: float = 0.0):
    """QKV projections of both streams into ``output`` [Ma + Mb, 3*H*128] (rows: stream a, then stream b) with one
    GEMM launch; ``rotary_emb`` is the packed table of the concatenated token sequence.  Returns False when the
    layers cannot be grouped (nothing has been written)."""
    if not _pair_ok(proj_a, proj_b, xa, xb) or xb.shape[1] % 256:
        return False
    pool = ln_a[3] if ln_a is not None and len(ln_a) > 3 else None
    act, asc, lact, Ma = _quantize_pair(xa, proj_a, xb, proj_b, ln_a, ln_b, pool=pool)
    svdq_gemm_w4a4_cuda(
        act=act, wgt=proj_a.qweight, out=output, ascales=asc, wscales=proj_a.wscales, lora_act_in=lact, lora_up=proj_a.proj_up,
        bias=proj_a.bias, norm_q=nq_a.weight, norm_k=nk_a.weight, rotary_emb=rotary_emb.reshape(-1, rotary_emb.shape[-1]),
        out_vt=out_vt, lora_scales=getattr(proj_a, "lora_scales", None), q_scale=q_scale,
        second=_second(proj_b, norm_q=nq_b.weight, norm_k=nk_b.weight), split_rows=Ma)
    return True


def quantize_two(x, lin_a, lin_b, ln=None):
    """The SAME input quantised for two layers (different smoothing / low-rank factors) in one launch: the grouped
    quantiser with both streams reading ``x``; returns ``((codes, scales, lora_act) for lin_a, ... for lin_b)`` as views
    of one set of buffers, or None when the shapes do not allow it (rows not a multiple of 256, different ranks)."""
    M, K = x.shape[-2], x.shape[-1]
    if x.numel() != M * K or M % 256 or lin_a.in_features != lin_b.in_features or lin_a.rank != lin_b.rank:
        return None
    lin_a._ensure_layout()
    lin_b._ensure_layout()
    R, dev = lin_a.rank, x.device
    pool = ln[3] if ln is not None and len(ln) > 3 else None
    act = torch.empty(2 * M, K * 3 // 4, dtype=torch.uint8, device=dev)
    asc = torch.empty(K // 64, 2 * M, dtype=x.dtype, device=dev)
    lact, zeroed = alloc_lora_act(2 * M, R, dev, pool)
    sec = dict(input=x, smooth=lin_b.smooth_factor, lora_down=lin_b.proj_down)
    if ln is not None:
        sec.update(ln_stats=ln[0], mod_scale=ln[1], mod_shift=ln[2])
    svdq_quantize_w4a4_act_fuse_lora_cuda(x.reshape(M, K), output=act, oscales=asc, lora_down=lin_a.proj_down, lora_act_out=lact,
                                          smooth=lin_a.smooth_factor, ln=None if ln is None else ln[:3], lora_act_zeroed=zeroed, second=sec)
    ascf, n = asc.view(-1), M * (K // 64)
    return ((act[:M], ascf[:n].view(K // 64, M), lact[:M]), (act[M:], ascf[n:].view(K // 64, M), lact[M:]))
