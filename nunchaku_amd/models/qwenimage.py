"""Qwen-Image transformer block and model on the SVDQuant hot path (BASELINE config 5).

Reference: ``NunchakuQwenImageTransformerBlock`` / ``NunchakuQwenImageTransformer2DModel``
(nunchaku/models/transformers/transformer_qwenimage.py:159-307, 309-560), ``NunchakuQwenAttention`` (:37-157) with
``NunchakuQwenImageNaiveFA2Processor`` (models/attention_processors/qwenimage.py), ``NunchakuFeedForward``
(models/attention.py:76-123) and ``CPUOffloadManager`` (models/utils.py:52-262).  Module and parameter NAMES follow the
reference (``img_mod.1``, ``attn.to_qkv``, ``attn.to_out.0``, ``attn.add_qkv_proj``, ``attn.to_add_out``, ``img_mlp.net.0.proj``,
``img_mlp.net.2``, ...), so a reference checkpoint's state dict loads key for key.

Dual-stream block, hidden 3072 = 24 heads x 128, MLP x4, every projection an ``SVDQW4A4Linear``, the two modulation
projections ``AWQW4A16Linear`` (W4A16 GEMV).  What differs from the reference's op sequence is WHERE the attention glue runs:
the reference projects QKV with a plain quantised linear and applies RMSNorm(q, k), the complex rotary multiplication and
SDPA as separate torch / diffusers ops; Qwen's rotary (``apply_rotary_emb_qwen(use_real=False)``: consecutive channel pairs
times a unit complex number) is exactly the adjacent-pair rotation of the FLUX QKV epilogue, so here the projection runs
with the fused RMSNorm + RoPE epilogue (V written transposed) and attention is this library's kernel on the QKV buffer in
place -- both streams in ONE launch each when the text length is a multiple of 256 (grouped launches).  ``fused_qkv = False``
on the attention module selects the reference's op-for-op sequence (plain projection, torch RMSNorm / rotary, SDPA).
"""

from __future__ import annotations

import json
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F
from torch import nn

from ..ops.attention import attention_packed, attention_packed_quantized, kv_valid_ranges, q_prescale
from ..ops.elementwise import residual_gate_stats, residual_gate_stats_pair
from ..ops.fused import (fused_gelu_mlp, fused_gelu_mlp_pair, fused_qkv_norm_rottary, fused_qkv_norm_rottary_pair, linear_pair,
                         linear_pair_quantized)
from ..ops.gemv import awq_gemv_w4a16_batched, awq_gemv_w4a16_cuda
from ..utils import pad_tensor
from .embeddings import pack_rotemb
from .linear import AWQW4A16Linear, SVDQW4A4Linear, synthetic_codes_
from .offload import CPUOffloadManager
from .transformer_flux import NunchakuModelLoaderMixin


def _pad256(n: int) -> int:
    return (n + 255) // 256 * 256


class _GELUProj(nn.Module):
    """``net.0`` of a diffusers FeedForward with ``activation_fn="gelu-approximate"``: holds ``proj`` (the activation itself
    is fused into the projection's GEMM epilogue)."""

    def __init__(self, dim, hidden, kw):
        super().__init__()
        self.proj = SVDQW4A4Linear(dim, hidden, **kw)


class NunchakuFeedForward(nn.Module):
    """reference: models/attention.py:76-123 -- ``net = [GELU(proj), Dropout, Linear]``; fc1 -> GELU -> fc2 with the
    requantisation fused into fc1's epilogue."""

    def __init__(self, dim, kw, mult: int = 4):
        super().__init__()
        self.net = nn.ModuleList([_GELUProj(dim, mult * dim, kw), nn.Identity(),
                                  SVDQW4A4Linear(mult * dim, dim, **{**kw, "act_unsigned": True})])

    def forward(self, x):
        return fused_gelu_mlp(x, self.net[0].proj, self.net[2])


class NunchakuQwenAttention(nn.Module):
    """Joint attention of a Qwen-Image block (reference :37-157): fused QKV projections per stream, QK RMSNorm, rotary,
    joint softmax attention over [text; image], output projections per stream."""

    fused_qkv = True  # False: the reference's op-for-op sequence (processor NunchakuQwenImageNaiveFA2Processor)

    def __init__(self, dim, heads, kw):
        super().__init__()
        self.heads, self.head_dim = heads, dim // heads
        dt, dev = kw["torch_dtype"], kw["device"]
        self.to_qkv = SVDQW4A4Linear(dim, 3 * dim, **kw)
        self.add_qkv_proj = SVDQW4A4Linear(dim, 3 * dim, **kw)
        self.to_out = nn.ModuleList([SVDQW4A4Linear(dim, dim, **kw), nn.Identity()])  # diffusers: [Linear, Dropout]
        self.to_add_out = SVDQW4A4Linear(dim, dim, **kw)
        self.norm_q = nn.RMSNorm(self.head_dim, eps=1e-6, dtype=dt, device=dev)
        self.norm_k = nn.RMSNorm(self.head_dim, eps=1e-6, dtype=dt, device=dev)
        self.norm_added_q = nn.RMSNorm(self.head_dim, eps=1e-6, dtype=dt, device=dev)
        self.norm_added_k = nn.RMSNorm(self.head_dim, eps=1e-6, dtype=dt, device=dev)
        self.added_kv_proj_dim = dim

    def forward(self, hidden_states, encoder_hidden_states, encoder_hidden_states_mask=None, attention_mask=None,
                image_rotary_emb=None, kv_valid=None, **kwargs):
        """-> (image stream output, text stream output), as the reference's processor returns them.
        ``image_rotary_emb`` = (img_freqs, txt_freqs) complex ``[T, 64]`` (diffusers ``QwenEmbedRope``) or, precomputed
        once per forward by the model, the packed real tables of :func:`pack_qwen_rotary`."""
        if attention_mask is not None:
            raise NotImplementedError("attention_mask is not supported")
        B, t_img, _ = hidden_states.shape
        t_txt = encoder_hidden_states.shape[1]
        tokens, hd = t_txt + t_img, self.heads * self.head_dim
        packed = image_rotary_emb if isinstance(image_rotary_emb, dict) else pack_qwen_rotary(*image_rotary_emb)
        if self.fused_qkv and B == 1 and self.head_dim == 128 and tokens % 128 == 0:
            qkv = torch.empty(tokens, 3 * hd, dtype=hidden_states.dtype, device=hidden_states.device)
            vt = torch.empty(hd, tokens, dtype=hidden_states.dtype, device=hidden_states.device)
            qs = q_prescale(self.head_dim)  # Q leaves the QKV GEMM times scale * log2(e): the attention kernel's fast geometry
            done = False
            if t_txt % 256 == 0 and t_img % 256 == 0:  # both streams in one launch (rows: text first)
                done = fused_qkv_norm_rottary_pair(encoder_hidden_states, self.add_qkv_proj, self.norm_added_q, self.norm_added_k,
                                                   hidden_states, self.to_qkv, self.norm_q, self.norm_k, packed["all"], qkv, out_vt=vt, q_scale=qs)
            if not done:
                fused_qkv_norm_rottary(encoder_hidden_states, self.add_qkv_proj, self.norm_added_q, self.norm_added_k, packed["txt"],
                                       output=qkv[:t_txt], out_vt=vt[:, :t_txt], q_scale=qs)
                fused_qkv_norm_rottary(hidden_states, self.to_qkv, self.norm_q, self.norm_k, packed["img"], output=qkv[t_txt:],
                                       out_vt=vt[:, t_txt:], q_scale=qs)
            o = attention_packed(qkv, vt, self.heads, q_prescaled=True, kv_valid=kv_valid).unsqueeze(0)
        else:
            if kv_valid is not None:
                raise RuntimeError("NunchakuQwenAttention: padded token streams need the fused QKV path (fused_qkv, batch 1, head_dim 128)")
            o = self._reference_ops(hidden_states, encoder_hidden_states, packed)
        if t_txt % 256 == 0 and B == 1:
            txt, img = linear_pair(o[:, :t_txt], self.to_add_out, o[:, t_txt:], self.to_out[0])
            return img, txt
        return self.to_out[0](o[:, t_txt:]), self.to_add_out(o[:, :t_txt])

    def forward_fused_norm(self, hidden, enc, packed, ln_img, ln_txt, kv_valid=None):
        """The block's fused path (``NunchakuQwenImageTransformerBlock.forward_fused``): ``hidden`` / ``enc`` are the UN-normalised
        streams, LayerNorm + modulation run inside the QKV quantiser (``ln_* = (stats, scale incl. +1, shift, ZeroPool)``), both
        streams share every launch, the attention epilogue emits the output projections' quantised input.  B = 1, token counts
        multiples of 256.  -> (image stream output, text stream output)."""
        t_txt, t_img = enc.shape[1], hidden.shape[1]
        tokens, hd = t_txt + t_img, self.heads * self.head_dim
        qkv = torch.empty(tokens, 3 * hd, dtype=hidden.dtype, device=hidden.device)
        vt = torch.empty(hd, tokens, dtype=hidden.dtype, device=hidden.device)
        ok = fused_qkv_norm_rottary_pair(enc, self.add_qkv_proj, self.norm_added_q, self.norm_added_k, hidden, self.to_qkv, self.norm_q,
                                         self.norm_k, packed["all"], qkv, out_vt=vt, ln_a=ln_txt, ln_b=ln_img, q_scale=q_prescale(self.head_dim))
        if not ok:
            raise RuntimeError("forward_fused_norm: the two streams' projections cannot share a launch (shapes / ranks differ)")
        qres = attention_packed_quantized(qkv, vt, self.heads, self.to_out[0], lin_first=self.to_add_out, split_rows=t_txt, pool=ln_txt[3],
                                          q_prescaled=True, kv_valid=kv_valid)
        if qres is not None:
            txt, img = linear_pair_quantized(*qres, self.to_add_out, self.to_out[0], t_txt)
            return img, txt
        # (shapes the attention epilogue's quantiser does not take -- the two projections' ranks differ, e.g. a runtime LoRA on one of them: the 16-bit
        #  round trip; Q left the QKV GEMM prescaled all the same)
        o = attention_packed(qkv, vt, self.heads, q_prescaled=True, kv_valid=kv_valid).unsqueeze(0)
        txt, img = linear_pair(o[:, :t_txt], self.to_add_out, o[:, t_txt:], self.to_out[0])
        return img, txt

    def _reference_ops(self, hidden_states, encoder_hidden_states, packed):
        """NunchakuQwenImageNaiveFA2Processor, op for op: plain quantised projections, torch RMSNorm, rotary as the complex
        product in fp32 with one rounding, ``scaled_dot_product_attention`` over [text; image]."""
        B = hidden_states.shape[0]
        shp = (B, -1, self.heads, self.head_dim)

        def stream(x, proj, nq, nk, cs):
            q, k, v = (t.view(shp) for t in proj(x).chunk(3, dim=-1))
            return _rotate(nq(q), cs), _rotate(nk(k), cs), v

        tq, tk, tv = stream(encoder_hidden_states, self.add_qkv_proj, self.norm_added_q, self.norm_added_k, packed["txt_cs"])
        iq, ik, iv = stream(hidden_states, self.to_qkv, self.norm_q, self.norm_k, packed["img_cs"])
        q, k, v = (torch.cat(p, dim=1).transpose(1, 2) for p in ((tq, iq), (tk, ik), (tv, iv)))
        o = F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=False)
        return o.transpose(1, 2).flatten(2, 3).to(hidden_states.dtype)


def _rotate(x: torch.Tensor, cs: torch.Tensor) -> torch.Tensor:
    """``apply_rotary_emb_qwen(x, freqs, use_real=False)``: consecutive channel pairs times cos + i sin, fp32, one rounding.
    x [B, T, H, D]; cs [T, D/2, 2] = (cos, sin)."""
    xf = x.float().unflatten(-1, (-1, 2))
    c, s = cs[None, :, None, :, 0], cs[None, :, None, :, 1]
    out = torch.stack([xf[..., 0] * c - xf[..., 1] * s, xf[..., 0] * s + xf[..., 1] * c], dim=-1)
    return out.flatten(-2).to(x.dtype)


def pack_qwen_rotary(img_freqs: torch.Tensor, txt_freqs: torch.Tensor) -> dict:
    """Complex rotary tables ``[T, 64]`` of the image and the text stream (diffusers ``QwenEmbedRope`` output) -> what the
    kernels read: the fused QKV epilogue's packed (sin, cos) tables (models/embeddings.py:pack_rotemb, 256-row padding)
    per stream and for the concatenated [text; image] sequence, plus plain (cos, sin) tables for the torch-op path."""
    def cs(f):
        return torch.stack([f.real.float(), f.imag.float()], dim=-1)  # [T, 64, (cos, sin)]

    def packed(c):
        sin_cos = torch.stack([c[..., 1], c[..., 0]], dim=-1)[None, :, :, None, :]  # [1, T, 64, 1, (sin, cos)]
        return pack_rotemb(pad_tensor(sin_cos, 256, 1))

    ic, tc = cs(img_freqs), cs(txt_freqs)
    # "all": the joint [text | image] sequence with EVERY stream on a 256-row boundary (zero entries for the padding rows in between)
    tc_pad = F.pad(tc, (0, 0, 0, 0, 0, -tc.shape[0] % 256))
    return {"img": packed(ic), "txt": packed(tc), "all": packed(torch.cat([tc_pad, ic], dim=0)), "img_cs": ic, "txt_cs": tc}


def qwen_rope_freqs(img_shape: tuple[int, int, int], txt_len: int, axes_dim=(16, 56, 56), theta: float = 10000.0,
                    scale_rope: bool = True, device="cpu"):
    """Rotary frequencies of one (frames, height, width) latent grid and ``txt_len`` text tokens, restated from diffusers'
    ``QwenEmbedRope`` (transformer_qwenimage.py; diffusers is not installed here): per axis ``exp(i * pos * theta^(-2j/d))``,
    image positions centred per axis when ``scale_rope``, text positions continuing after ``max(height, width) / 2``.
    -> (img_freqs [F*H*W, 64] complex64, txt_freqs [txt_len, 64] complex64)."""
    def axis(pos, d):
        inv = 1.0 / theta ** (torch.arange(0, d, 2, dtype=torch.float32, device=device) / d)
        return torch.polar(torch.ones(len(pos), d // 2, device=device), pos.float()[:, None] * inv[None])

    frame, height, width = img_shape

    def centred(n):
        return torch.cat([torch.arange(-(n - n // 2), 0, device=device), torch.arange(0, n // 2, device=device)]) if scale_rope \
            else torch.arange(n, device=device)

    ff = axis(torch.arange(frame, device=device), axes_dim[0])[:, None, None, :].expand(frame, height, width, -1)
    fh = axis(centred(height), axes_dim[1])[None, :, None, :].expand(frame, height, width, -1)
    fw = axis(centred(width), axes_dim[2])[None, None, :, :].expand(frame, height, width, -1)
    img = torch.cat([ff, fh, fw], dim=-1).reshape(frame * height * width, -1)
    start = max(height // 2, width // 2) if scale_rope else max(height, width)
    tpos = torch.arange(start, start + txt_len, device=device)
    txt = torch.cat([axis(tpos, d) for d in axes_dim], dim=-1)
    return img, txt


class NunchakuQwenImageTransformerBlock(nn.Module):
    """reference :159-307.  ``scale_shift`` = 1.0: the checkpoint's modulation does NOT carry the +1 of the scale (unlike
    nunchaku's FLUX checkpoints), it is added here (:203-205)."""

    def __init__(self, dim: int = 3072, num_attention_heads: int = 24, attention_head_dim: int = 128, rank: int = 32,
                 scale_shift: float = 1.0, torch_dtype: torch.dtype = torch.bfloat16, device="cuda"):
        super().__init__()
        assert dim == num_attention_heads * attention_head_dim
        kw = dict(rank=rank, torch_dtype=torch_dtype, device=device)
        self.dim = dim
        self.img_mod = nn.Sequential(nn.SiLU(), AWQW4A16Linear(dim, 6 * dim, torch_dtype=torch_dtype, device=device))
        self.img_norm1 = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.attn = NunchakuQwenAttention(dim, num_attention_heads, kw)
        self.img_norm2 = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.img_mlp = NunchakuFeedForward(dim, kw)
        self.txt_mod = nn.Sequential(nn.SiLU(), AWQW4A16Linear(dim, 6 * dim, torch_dtype=torch_dtype, device=device))
        self.txt_norm1 = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.txt_norm2 = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.txt_mlp = NunchakuFeedForward(dim, kw)
        self.scale_shift = scale_shift

    def _modulate(self, x, mod_params):
        shift, scale, gate = mod_params.chunk(3, dim=-1)
        if self.scale_shift != 0:
            scale = scale + self.scale_shift
        return x * scale.unsqueeze(1) + shift.unsqueeze(1), gate.unsqueeze(1)

    def modulation(self, temb_act):
        """Both streams' modulation vectors, de-interleaved by the GEMV itself: (img [6, dim], txt [6, dim]) with rows
        shift1, scale1 (+1 included), gate1, shift2, scale2 (+1), gate2 -- what :meth:`forward_fused` takes as ``mods``."""
        outs = []
        for lin in (self.img_mod[1], self.txt_mod[1]):
            m = awq_gemv_w4a16_cuda(temb_act, lin.qweight, lin.wscales, lin.wzeros, 1, lin.out_features, lin.in_features, lin.group_size,
                                    lin.bias, out_chunks=6).view(6, -1)
            if self.scale_shift != 0:
                m[1::3] += self.scale_shift  # a 16-bit add, as the reference's `scale + scale_shift` (:203-205)
            outs.append(m)
        return outs

    def forward_fused(self, hidden, enc, temb_act, packed_rot, stats, mods=None, kv_valid=None):
        """The block on this library's fused passes (B = 1, token counts multiples of 256): LayerNorm + modulation inside the
        quantisers, gated residual + the next LayerNorm's statistics in one element-wise pass per stage (both streams per launch),
        grouped GEMM launches, attention-side quantiser.  Same 16-bit rounding points as :meth:`forward`'s torch ops (the fused
        passes are bit-exact restatements of them).  ``stats`` = ((txt stats, ZeroPool), img stats) of the
        block's inputs; returns (enc, hidden, stats of the outputs)."""
        (e_stats, pool), h_stats = stats
        im, tm = mods if mods is not None else self.modulation(temb_act)
        att = self.attn
        img_a, txt_a = att.forward_fused_norm(hidden, enc, packed_rot, ln_img=(h_stats, im[1], im[0]), ln_txt=(e_stats, tm[1], tm[0], pool),
                                              kv_valid=kv_valid)
        mp = _pad256(hidden.shape[1]) + _pad256(enc.shape[1])
        r_mlp = self.img_mlp.net[0].proj.rank + self.img_mlp.net[2].rank
        enc, e_stats, hidden, h_stats, pool = residual_gate_stats_pair(enc, txt_a, tm[2], hidden, img_a, im[2], zero_floats=mp * r_mlp)
        txt_f, img_f = fused_gelu_mlp_pair(enc, self.txt_mlp.net[0].proj, self.txt_mlp.net[2], hidden, self.img_mlp.net[0].proj, self.img_mlp.net[2],
                                           ln_a=(e_stats, tm[4], tm[3], pool), ln_b=(h_stats, im[4], im[3]))
        fp16 = hidden.dtype == torch.float16  # the reference clips both streams at the end of an fp16 block (:300-303)
        enc, e_stats, hidden, h_stats, pool = residual_gate_stats_pair(
            enc, txt_f, tm[5], hidden, img_f, im[5], zero_floats=mp * (att.to_qkv.rank + att.to_out[0].rank), clamp_fp16_a=fp16, clamp_fp16_b=fp16)
        return enc, hidden, ((e_stats, pool), h_stats)

    def forward(self, hidden_states, encoder_hidden_states, encoder_hidden_states_mask=None, temb=None, image_rotary_emb=None,
                joint_attention_kwargs=None, kv_valid=None):
        B = temb.shape[0]
        # nunchaku's modulation weights are stored channel-interleaved: [B, dim * 6] -> [B, 6 * dim] (:236-243)
        img_mod = self.img_mod(temb).view(B, -1, 6).transpose(1, 2).reshape(B, -1)
        txt_mod = self.txt_mod(temb).view(B, -1, 6).transpose(1, 2).reshape(B, -1)
        img_mod1, img_mod2 = img_mod.chunk(2, dim=-1)
        txt_mod1, txt_mod2 = txt_mod.chunk(2, dim=-1)
        img_x, img_gate1 = self._modulate(self.img_norm1(hidden_states), img_mod1)
        txt_x, txt_gate1 = self._modulate(self.txt_norm1(encoder_hidden_states), txt_mod1)
        img_attn, txt_attn = self.attn(hidden_states=img_x, encoder_hidden_states=txt_x, encoder_hidden_states_mask=encoder_hidden_states_mask,
                                       image_rotary_emb=image_rotary_emb, kv_valid=kv_valid, **(joint_attention_kwargs or {}))
        hidden_states = hidden_states + img_gate1 * img_attn
        encoder_hidden_states = encoder_hidden_states + txt_gate1 * txt_attn
        img_x2, img_gate2 = self._modulate(self.img_norm2(hidden_states), img_mod2)
        hidden_states = hidden_states + img_gate2 * self.img_mlp(img_x2)
        txt_x2, txt_gate2 = self._modulate(self.txt_norm2(encoder_hidden_states), txt_mod2)
        encoder_hidden_states = encoder_hidden_states + txt_gate2 * self.txt_mlp(txt_x2)
        if encoder_hidden_states.dtype == torch.float16:  # :300-303
            encoder_hidden_states = encoder_hidden_states.clip(-65504, 65504)
        if hidden_states.dtype == torch.float16:
            hidden_states = hidden_states.clip(-65504, 65504)
        return encoder_hidden_states, hidden_states


class _TimestepEmbed(nn.Module):
    """diffusers ``QwenTimestepProjEmbeddings``: sinusoidal(256, flip_sin_to_cos, scale 1000) -> Linear -> SiLU -> Linear."""

    def __init__(self, dim, dtype, device):
        super().__init__()
        self.timestep_embedder = nn.ModuleDict({"linear_1": nn.Linear(256, dim, dtype=dtype, device=device),
                                                "linear_2": nn.Linear(dim, dim, dtype=dtype, device=device)})

    def forward(self, timestep, dtype):
        half = 128
        freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=timestep.device) / half)
        args = timestep.float()[:, None] * 1000.0 * freqs[None]
        emb = torch.cat([args.cos(), args.sin()], dim=-1).to(dtype)
        return self.timestep_embedder["linear_2"](F.silu(self.timestep_embedder["linear_1"](emb)))


try:  # guarded, as transformer_flux.py: a real diffusers subclass when diffusers is importable
    from diffusers import QwenImageTransformer2DModel as _DiffusersQwen

    HAVE_DIFFUSERS_QWEN = True
except Exception:
    _DiffusersQwen = None
    HAVE_DIFFUSERS_QWEN = False

_QWEN_DEFAULT_CONFIG = dict(num_layers=60, num_attention_heads=24, attention_head_dim=128, in_channels=64, out_channels=16,
                            joint_attention_dim=3584, patch_size=2, axes_dims_rope=(16, 56, 56), guidance_embeds=False)


class NunchakuQwenImageTransformer2DModel(_DiffusersQwen if HAVE_DIFFUSERS_QWEN else nn.Module, NunchakuModelLoaderMixin):
    """reference :309-560 (a ``diffusers.QwenImageTransformer2DModel`` subclass + ``NunchakuModelLoaderMixin``; here the same
    when diffusers is importable -- skeleton on the meta device, every parametrised sub-module replaced under its diffusers
    name by ``_patch_model`` -- and a plain ``nn.Module`` with the pipeline's call contract otherwise): 60 dual-stream blocks,
    ``from_pretrained`` for nunchaku ``.safetensors`` checkpoints, ``set_offload`` for layer-wise host offload."""

    # True (and B == 1, token counts multiples of 256): the blocks run on the fused AdaLayerNormZero / residual passes and grouped
    # launches (NunchakuQwenImageTransformerBlock.forward_fused); False: the reference's torch-op sequence per block
    fused_norm = True
    # True: all modulation GEMVs of a step in one batched launch (resident models only)
    batched_mods = True
    # True: both token streams are padded to 256 rows inside forward(), so that every token count runs the fused path (False: A/B)
    padded_tokens = True

    def __init__(self, num_layers: int = 60, num_attention_heads: int = 24, attention_head_dim: int = 128, in_channels: int = 64,
                 out_channels: int = 16, joint_attention_dim: int = 3584, patch_size: int = 2, axes_dims_rope=(16, 56, 56),
                 rank: int = 32, torch_dtype: torch.dtype = torch.bfloat16, device="cuda", guidance_embeds: bool = False,
                 _patch: bool = True, **kwargs):
        self.offload = kwargs.pop("offload", False) and False  # set_offload() switches it on once the weights are loaded
        self.offload_manager = None
        cfg = dict(num_layers=num_layers, num_attention_heads=num_attention_heads, attention_head_dim=attention_head_dim,
                   in_channels=in_channels, out_channels=out_channels, joint_attention_dim=joint_attention_dim, patch_size=patch_size,
                   axes_dims_rope=tuple(axes_dims_rope), guidance_embeds=guidance_embeds)
        if HAVE_DIFFUSERS_QWEN:
            with torch.device("meta"):
                super().__init__(**cfg, **kwargs)
        else:
            nn.Module.__init__(self)
            self.config = SimpleNamespace(**cfg)
        self.offload, self.offload_manager = False, None
        if _patch:
            self._patch_model(rank=rank, torch_dtype=torch_dtype, device=device)

    def _patch_model(self, rank: int = 32, torch_dtype: torch.dtype = torch.bfloat16, device="cuda", **kwargs):
        """reference :339-356 -- the blocks (and here every other parametrised module: the skeleton is on the meta device)."""
        if kwargs.get("precision", "int4") not in ("int4",):
            raise NotImplementedError("NVFP4 checkpoints need Blackwell's block-scaled mma; use the int4 checkpoint on MI355X")
        c = self.config
        get = (lambda k: c[k]) if hasattr(c, "keys") else (lambda k: getattr(c, k))
        heads, hd = get("num_attention_heads"), get("attention_head_dim")
        dim = heads * hd
        self.inner_dim, self.axes = dim, tuple(get("axes_dims_rope"))
        self.time_text_embed = _TimestepEmbed(dim, torch_dtype, device)
        self.txt_norm = nn.RMSNorm(get("joint_attention_dim"), eps=1e-6, dtype=torch_dtype, device=device)
        self.img_in = nn.Linear(get("in_channels"), dim, dtype=torch_dtype, device=device)
        self.txt_in = nn.Linear(get("joint_attention_dim"), dim, dtype=torch_dtype, device=device)
        self.transformer_blocks = nn.ModuleList([
            NunchakuQwenImageTransformerBlock(dim, heads, hd, rank=rank, torch_dtype=torch_dtype, device=device)
            for _ in range(get("num_layers"))])
        self.norm_out = nn.ModuleDict({"linear": nn.Linear(dim, 2 * dim, dtype=torch_dtype, device=device)})  # AdaLayerNormContinuous
        self.proj_out = nn.Linear(dim, get("patch_size") ** 2 * get("out_channels"), dtype=torch_dtype, device=device)
        if hasattr(self, "pos_embed"):
            self.pos_embed = nn.Identity()  # rotary tables: qwen_rope_freqs (restated QwenEmbedRope)
        self.dtype_ = torch_dtype
        self._is_initialized = True
        return self

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, **kwargs):
        """A nunchaku Qwen-Image ``.safetensors`` file (reference :358-413): the checkpoint's keys ARE the module names of this
        class (``transformer_blocks.N.attn.to_qkv.qweight`` ...), NVFP4-only tensors are dropped, ``offload=True`` switches the
        layer-wise host offload on after loading."""
        sd, meta = cls._read_safetensors(pretrained_model_name_or_path)
        config = dict(_QWEN_DEFAULT_CONFIG)
        config.update({k: v for k, v in json.loads(meta.get("config", "{}")).items() if k in _QWEN_DEFAULT_CONFIG})
        qcfg = json.loads(meta.get("quantization_config", "{}"))
        if any(k.endswith(".wcscales") for k in sd) or qcfg.get("weight", {}).get("dtype", "int4") not in ("int4",):
            raise NotImplementedError("NVFP4 checkpoints need Blackwell's block-scaled mma; use the int4 checkpoint on MI355X")
        model = cls(**config, rank=qcfg.get("rank", 32), torch_dtype=kwargs.get("torch_dtype", torch.bfloat16), device=kwargs.get("device", "cuda"))
        sd = {k: v for k, v in sd.items() if k.rsplit(".", 1)[-1] not in ("wtscale", "wcscales")}  # patch_scale_key (utils.py:151-173)
        own = dict(model.named_parameters())
        for k, v in sd.items():
            if k in own and own[k].dtype != v.dtype:
                raise TypeError(f"{k}: checkpoint dtype {v.dtype} != model dtype {own[k].dtype}")
        model.load_state_dict(sd)
        if kwargs.get("offload", False):
            model.set_offload(True, **{k: kwargs[k] for k in ("num_blocks_on_gpu", "use_pin_memory", "num_slots") if k in kwargs})
        return model

    if not HAVE_DIFFUSERS_QWEN:  # ModelMixin provides these

        @property
        def dtype(self):
            return self.dtype_

        @property
        def device(self):
            return self.proj_out.weight.device

    def svdq_layers(self):
        return [m for m in self.modules() if isinstance(m, SVDQW4A4Linear)]

    def set_offload(self, offload: bool, **kwargs):
        """reference :415-451 -- layer-wise host offload of the transformer blocks (models/offload.py)."""
        if offload == self.offload:
            return
        self.offload = offload
        if offload:
            self.offload_manager = CPUOffloadManager(
                list(self.transformer_blocks), device=kwargs.get("device", self.device), use_pin_memory=kwargs.get("use_pin_memory", True),
                on_gpu_modules=[self.img_in, self.txt_in, self.txt_norm, self.time_text_embed, self.norm_out, self.proj_out],
                num_blocks_on_gpu=kwargs.get("num_blocks_on_gpu", 1),
                # four device slots (the reference ping-pongs between two buffers, models/utils.py:188-221): measured 239.9 / 228.0 / 220.5 ms per 1024^2
                # step with 2 / 3 / 4 slots, 237 with 6 or 8 (profiles/r5_qwen_offload_ring_depth.txt): three loads in flight keep the link busier
                num_slots=kwargs.get("num_slots", 4))
        else:
            # the blocks are complete modules on pinned host memory: bring them back (the reference leaves them on the CPU and
            # relies on a later .to(device); a model that has just been told "no offload" should simply run)
            self.offload_manager.restore()
            self.offload_manager = None
            torch.cuda.empty_cache()

    def to(self, *args, **kwargs):
        """reference :560-612: no dtype casts of a quantised model; with offload on, the blocks stay where the manager put them."""
        if any(isinstance(a, torch.dtype) for a in args) or "dtype" in kwargs:
            raise ValueError("Casting a quantized model to a new `dtype` is unsupported")
        if self.offload:
            return self
        return super().to(*args, **kwargs)

    @torch.no_grad()
    def init_synthetic_(self, seed: int = 0, codes: str = "uniform"):
        """Random-init weights of the Qwen-Image shape (no checkpoints in this environment), written in the checkpoint layout
        (``codes``: models/linear.py ``synthetic_codes_``)."""
        dev = self.proj_out.weight.device
        g = torch.Generator(device=dev).manual_seed(seed)
        for m in self.modules():
            if isinstance(m, SVDQW4A4Linear):
                K = m.in_features
                synthetic_codes_(m.qweight, m.wscales, K, g, codes)
                if m.bias is not None:
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g, device=dev) * 0.02)
                m.smooth_factor.copy_(torch.rand((K,), generator=g, device=dev) + 0.5)
                m.smooth_factor_orig.copy_(m.smooth_factor)
                m.proj_down.copy_(torch.randn(m.proj_down.shape, generator=g, device=dev) * (0.5 / math.sqrt(K)))
                m.proj_up.copy_(torch.randn(m.proj_up.shape, generator=g, device=dev) * (0.5 / math.sqrt(m.rank)))
                m._amd_layout = False
            elif isinstance(m, AWQW4A16Linear):
                sc = 1.0 / (4.6 * math.sqrt(m.in_features))
                m.qweight.copy_(torch.randint(-2 ** 31, 2 ** 31, m.qweight.shape, generator=g, device=dev, dtype=torch.int64))
                m.wscales.copy_((torch.rand(m.wscales.shape, generator=g, device=dev) * 0.5 + 0.75) * sc)
                m.wzeros.copy_(m.wscales.float() * -7.5)
                m.bias.zero_()
            elif isinstance(m, nn.Linear):
                m.weight.copy_(torch.randn(m.weight.shape, generator=g, device=dev) / math.sqrt(m.in_features))
                m.bias.zero_()
            elif isinstance(m, nn.RMSNorm):
                m.weight.fill_(1.0)
        return self

    def forward(self, hidden_states, encoder_hidden_states=None, encoder_hidden_states_mask=None, timestep=None, img_shapes=None,
                txt_seq_lens=None, guidance=None, attention_kwargs=None, controlnet_block_samples=None, return_dict: bool = True):
        """hidden_states [1, T_img, 64] (packed 2x2 latent patches); encoder_hidden_states [1, T_txt, 3584]; timestep [1] (already
        divided by 1000 by the pipeline); img_shapes [(frames, H/2, W/2)] of the latent grid.  -> [1, T_img, 64]."""
        dt = self.dtype_
        hidden = self.img_in(hidden_states)
        enc = self.txt_in(self.txt_norm(encoder_hidden_states))
        temb = self.time_text_embed(timestep.to(dt), dt)
        t_txt, t_img = enc.shape[1], hidden.shape[1]
        shape = img_shapes[0] if img_shapes else (1, int(math.isqrt(hidden.shape[1])), int(math.isqrt(hidden.shape[1])))
        if isinstance(shape, (list, tuple)) and isinstance(shape[0], (list, tuple)):
            shape = shape[0]
        # the rotary tables depend on the grid, the text length and the device only: built once per (shape, length) and kept (one entry) --
        # except while the stream is capturing: then they are computed inside the graph, whose pool owns them (a graph must not point at
        # tables that only this cache keeps alive: the next eager call with another grid would free them under it; models/flux.py)
        rkey = (tuple(shape), t_txt, str(hidden.device))
        use_cache = not (hidden.is_cuda and torch.cuda.is_current_stream_capturing())
        cached = getattr(self, "_rot_cache", None) if use_cache else None
        if cached is not None and cached[0] == rkey:
            rot = cached[1]
        else:
            rot = pack_qwen_rotary(*qwen_rope_freqs(tuple(shape), t_txt, self.axes, device=hidden.device))
            if use_cache:
                self._rot_cache = (rkey, rot)
        compute_stream = torch.cuda.current_stream()
        if self.offload:
            self.offload_manager.initialize(compute_stream)
        # EVERY token count runs the hot path (the reference pads any M to 256 rows, Linear.cpp:445-446; its own quality gate is 1664 x 928 =
        # 6032 image tokens, tests/v1/qwenimage/test_qwenimage.py:21,118, and a prompt's text length is arbitrary): both streams are padded to
        # 256 rows with zero tokens behind the embedders, the attention kernel masks the padded keys (kv_valid), the real image rows are
        # sliced out at the end.  `encoder_hidden_states_mask` is accepted and ignored, as the reference's processor ignores it
        # (attention_processors/qwenimage.py: the batch-1 pipeline passes an all-ones mask).
        hot = hidden.shape[0] == 1 and NunchakuQwenAttention.fused_qkv and self.transformer_blocks[0].attn.head_dim == 128 and not attention_kwargs
        kv_valid = kv_valid_ranges(t_txt, t_img) if hot and self.padded_tokens else None
        if kv_valid is not None:
            enc, hidden = F.pad(enc, (0, 0, 0, -t_txt % 256)), F.pad(hidden, (0, 0, 0, -t_img % 256))
        fused = self.fused_norm and hot and enc.shape[1] % 256 == 0 and hidden.shape[1] % 256 == 0
        stats = mods = temb_act = None
        if fused:
            temb_act = F.silu(temb)  # img_mod[0] / txt_mod[0] of every block: the same SiLU of the same embedding
            stats = ((residual_gate_stats(enc)[1], None), residual_gate_stats(hidden)[1])
            if not self.offload and self.batched_mods:
                # all 2 x num_layers modulation projections depend on temb only: ONE batched GEMV launch, +1 on the scale rows in two ops
                lins = [SimpleNamespace(qweight=l.qweight, wscales=l.wscales, wzeros=l.wzeros, bias=l.bias, out_features=l.out_features,
                                        in_features=l.in_features, group_size=l.group_size, out_chunks=6)
                        for b in self.transformer_blocks for l in (b.img_mod[1], b.txt_mod[1])]
                outs = awq_gemv_w4a16_batched(temb_act, lins)
                base = outs[0]._base if outs[0]._base is not None else outs[0]  # the launch's one output buffer: the layers' vectors back to back
                allm = base.reshape(-1)[: len(lins) * 6 * self.inner_dim].view(len(lins), 6, self.inner_dim)
                if self.transformer_blocks[0].scale_shift != 0:
                    allm[:, 1::3] += self.transformer_blocks[0].scale_shift
                mods = [(allm[2 * i], allm[2 * i + 1]) for i in range(len(self.transformer_blocks))]
        for i, block in enumerate(self.transformer_blocks):
            if self.offload:
                block = self.offload_manager.get_block(i)
            if fused:
                enc, hidden, stats = block.forward_fused(hidden, enc, temb_act, rot, stats, mods=None if mods is None else mods[i], kv_valid=kv_valid)
            else:
                enc, hidden = block(hidden_states=hidden, encoder_hidden_states=enc, encoder_hidden_states_mask=encoder_hidden_states_mask,
                                    temb=temb, image_rotary_emb=rot, joint_attention_kwargs=attention_kwargs, kv_valid=kv_valid)
            if controlnet_block_samples is not None:
                # the reference's (= diffusers') choice of the residual behind block i (transformer_qwenimage.py:546-550): one 16-bit add on the image
                # stream; the fused path needs the LayerNorm statistics of the sum -- the same pass.  Padded image rows get a zero residual.
                smp = controlnet_block_samples[i // -(-len(self.transformer_blocks) // len(controlnet_block_samples))].to(hidden.dtype)
                if smp.shape[1] != hidden.shape[1]:
                    smp = F.pad(smp, (0, 0, 0, hidden.shape[1] - smp.shape[1]))
                hidden, h_stats = residual_gate_stats(hidden, smp.contiguous(), want_stats=fused)
                if fused:
                    stats = (stats[0], h_stats)
            if self.offload:
                self.offload_manager.step(compute_stream)
        hidden = hidden[:, :t_img]
        scale, shift = self.norm_out["linear"](F.silu(temb)).chunk(2, dim=-1)  # AdaLayerNormContinuous
        hidden = F.layer_norm(hidden, (self.inner_dim,), eps=1e-6) * (1 + scale[:, None]) + shift[:, None]
        out = self.proj_out(hidden)
        if not return_dict:
            return (out,)
        from .transformer_flux import Transformer2DModelOutput

        return Transformer2DModelOutput(sample=out)
