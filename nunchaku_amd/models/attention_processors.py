"""Attention processors with the reference's call contract (nunchaku/models/attention_processors/flux.py:14-237):
``processor(attn, hidden_states, encoder_hidden_states=None, attention_mask=None, image_rotary_emb=...)`` on an attention
module that exposes ``to_qkv / norm_q / norm_k / to_out`` (and ``add_qkv_proj / norm_added_q / norm_added_k / to_add_out``
for a joint block), ``heads`` and ``head_dim``.

* ``NunchakuFluxFA2Processor``   -- fused QKV + RMSNorm + RoPE projection, then ``F.scaled_dot_product_attention``;
* ``NunchakuFluxFP16AttnProcessor`` -- the reference's "nunchaku-fp16" path through its own operator surface:
  ``fused_qkv_norm_rottary(..., output=(q, k, v), attn_tokens=...)`` + ``_C.ops.attention_fp16``.  On MI355X these two
  operators are adapters over ``svdq_attention`` (nunchaku_amd/_C.py); the copy-free form of the same idea is what
  ``FluxAttentionAMD`` runs by default (``out_vt`` side output + ``ops.attention`` on the QKV buffer in place).
"""

from __future__ import annotations

import math

import torch
from torch.nn import functional as F

from .._C import ops
from ..ops.fused import fused_qkv_norm_rottary


def _out_proj(attn, x):
    to_out = attn.to_out
    if isinstance(to_out, (torch.nn.ModuleList, torch.nn.Sequential, list, tuple)):  # diffusers: [Linear, Dropout]
        for m in to_out:
            x = m(x)
        return x
    return to_out(x)


def _is_joint(attn) -> bool:
    return getattr(attn, "added_kv_proj_dim", None) is not None or bool(getattr(attn, "joint", False))


class NunchakuFluxFA2Processor:
    """reference :14-111."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, image_rotary_emb=None, **kwargs):
        if attention_mask is not None:
            raise NotImplementedError("attention_mask is not supported")
        batch_size, _, channels = hidden_states.shape
        assert channels == attn.heads * attn.head_dim
        qkv = fused_qkv_norm_rottary(hidden_states, attn.to_qkv, attn.norm_q, attn.norm_k,
                                     image_rotary_emb[0] if isinstance(image_rotary_emb, tuple) else image_rotary_emb)
        if _is_joint(attn):
            assert encoder_hidden_states is not None and isinstance(image_rotary_emb, tuple)
            qkv_context = fused_qkv_norm_rottary(encoder_hidden_states, attn.add_qkv_proj, attn.norm_added_q, attn.norm_added_k,
                                                 image_rotary_emb[1])
            qkv = torch.cat([qkv_context, qkv], dim=1)
        shp = (batch_size, -1, attn.heads, attn.head_dim)
        q, k, v = (t.view(shp).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
        o = F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(batch_size, -1, attn.heads * attn.head_dim).to(q.dtype)
        if encoder_hidden_states is not None:
            t_txt = encoder_hidden_states.shape[1]
            return _out_proj(attn, o[:, t_txt:]), attn.to_add_out(o[:, :t_txt])
        return _out_proj(attn, o)


class NunchakuFluxFP16AttnProcessor:
    """reference :114-237 (``pad_size``: sequence padding of the packed buffers)."""

    def __init__(self, pad_size: int = 256):
        self.pad_size = pad_size

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, image_rotary_emb=None, **kwargs):
        pad = self.pad_size
        batch_size, _, channels = hidden_states.shape
        assert channels == attn.heads * attn.head_dim
        dev, dt = hidden_states.device, hidden_states.dtype

        def packed(n):
            return torch.empty(batch_size, attn.heads, n, attn.head_dim, dtype=dt, device=dev)

        if encoder_hidden_states is None:
            num_tokens = hidden_states.shape[1]
            num_tokens_pad = math.ceil(num_tokens / pad) * pad
            query, key, value = packed(num_tokens_pad), packed(num_tokens_pad), packed(num_tokens_pad)
            assert torch.is_tensor(image_rotary_emb)
            fused_qkv_norm_rottary(hidden_states, attn.to_qkv, attn.norm_q, attn.norm_k, image_rotary_emb,
                                   output=(query, key, value), attn_tokens=num_tokens)
        else:
            n_txt, n_img = encoder_hidden_states.shape[1], hidden_states.shape[1]
            n_txt_pad, n_img_pad = math.ceil(n_txt / pad) * pad, math.ceil(n_img / pad) * pad
            num_tokens_pad = n_txt_pad + n_img_pad
            query, key, value = packed(num_tokens_pad), packed(num_tokens_pad), packed(num_tokens_pad)
            assert isinstance(image_rotary_emb, tuple)
            fused_qkv_norm_rottary(hidden_states, attn.to_qkv, attn.norm_q, attn.norm_k, image_rotary_emb[0],
                                   output=(query[:, :, n_txt_pad:], key[:, :, n_txt_pad:], value[:, :, n_txt_pad:]), attn_tokens=n_img)
            fused_qkv_norm_rottary(encoder_hidden_states, attn.add_qkv_proj, attn.norm_added_q, attn.norm_added_k, image_rotary_emb[1],
                                   output=(query[:, :, :n_txt_pad], key[:, :, :n_txt_pad], value[:, :, :n_txt_pad]), attn_tokens=n_txt)
        out = torch.empty(batch_size, num_tokens_pad, attn.heads * attn.head_dim, dtype=dt, device=dev)
        ops.attention_fp16(query, key, value, out, attn.head_dim ** (-0.5))
        if encoder_hidden_states is None:
            return _out_proj(attn, out[:, :num_tokens])
        enc, hid = out[:, :n_txt], out[:, n_txt_pad:n_txt_pad + n_img]
        return _out_proj(attn, hid), attn.to_add_out(enc)
