"""FLUX.1-shaped denoising transformer built on the SVDQuant hot path (stand-alone).

Structure and call pattern follow the reference's V2 model
(nunchaku/models/transformers/transformer_flux_v2.py:118-342,430-561,
 nunchaku/models/attention_processors/flux.py:71-108, nunchaku/models/normalization.py:85-165,
 nunchaku/models/attention.py:76-123): 19 joint + 38 single blocks, hidden 3072 = 24 x 128,
MLP x4; every 3072-wide projection is an ``SVDQW4A4Linear`` driven through
``fused_qkv_norm_rottary`` / ``fused_gelu_mlp`` / ``forward``.

``diffusers`` is not a dependency: the few non-quantised pieces it would provide (embedders,
AdaLayerNorm modulation) are restated here with plain torch ops.  The AdaLN modulation projections are
AWQ W4A16 GEMVs (``AWQW4A16Linear``) and attention runs on this library's kernel, as in the reference
(SURVEY.md section 8f items 1 and 3).  Used by bench.py with synthetic weights and by the GPU tests; a
reference checkpoint's SVDQ tensors load into the ``SVDQW4A4Linear`` members unchanged.
"""

from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn

from ..ops.attention import attention_packed, attention_packed_quantized, kv_valid_ranges, q_prescale
from ..ops.elementwise import residual_gate_stats, residual_gate_stats_pair
from ..ops.gemv import awq_gemv_w4a16_batched
from ..ops.fused import (fused_gelu_mlp, fused_gelu_mlp_pair, fused_qkv_norm_rottary, fused_qkv_norm_rottary_pair,
                         linear_pair, linear_pair_quantized, quantize_two)
from ..utils import pad_tensor
from .embeddings import flux_pos_embed, pack_rotemb
from .linear import AWQW4A16Linear, SVDQW4A4Linear, synthetic_codes_


_FREQS: dict = {}


def timestep_embedding(t: torch.Tensor, dim: int = 256, max_period: float = 10000.0) -> torch.Tensor:
    """Sinusoidal embedding, (cos, sin) order, as diffusers' ``Timesteps(flip_sin_to_cos=True)``.  (The frequency table is a constant
    of (dim, period, device): built once -- three tiny launches less per call.)"""
    half = dim // 2
    key = (half, max_period, str(t.device))
    capturing = t.is_cuda and torch.cuda.is_current_stream_capturing()  # (a tensor made under capture lives in the graph's pool: not kept)
    freqs = None if capturing else _FREQS.get(key)
    if freqs is None:
        freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
        if not capturing:
            _FREQS[key] = freqs
    args = t.float()[:, None] * freqs[None]
    return torch.cat([args.cos(), args.sin()], dim=-1)


def _pad256(n: int) -> int:
    return (n + 255) // 256 * 256


def _pair_compatible(la, lb) -> bool:
    return (la.in_features == lb.in_features and la.out_features == lb.out_features and la.rank == lb.rank
            and (la.bias is None) == (lb.bias is None) and getattr(la, "lora_scales", None) == getattr(lb, "lora_scales", None))


class _MLPEmbedder(nn.Module):
    def __init__(self, d_in, d, dtype, device):
        super().__init__()
        self.linear_1 = nn.Linear(d_in, d, dtype=dtype, device=device)
        self.linear_2 = nn.Linear(d, d, dtype=dtype, device=device)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class _TimeTextEmbed(nn.Module):
    """diffusers' CombinedTimestep(Guidance)TextProjEmbeddings: ``timestep_embedder`` / ``guidance_embedder`` / ``text_embedder``,
    each ``linear_1 -> SiLU -> linear_2``."""

    def __init__(self, dim, pooled_dim, guidance, dtype, device):
        super().__init__()
        self.timestep_embedder = _MLPEmbedder(256, dim, dtype, device)
        self.guidance_embedder = _MLPEmbedder(256, dim, dtype, device) if guidance else None
        self.text_embedder = _MLPEmbedder(pooled_dim, dim, dtype, device)


class _AdaLNContinuous(nn.Module):
    """diffusers' AdaLayerNormContinuous of the output head: ``linear`` (dim -> 2 dim); the LayerNorm has no parameters."""

    def __init__(self, dim, dtype, device):
        super().__init__()
        self.linear = nn.Linear(dim, 2 * dim, dtype=dtype, device=device)


class FluxAttentionAMD(nn.Module):
    """Joint (img + txt) or single-stream attention with fused QKV/RMSNorm/RoPE projections."""

    def __init__(self, dim, heads, joint: bool, kw):
        super().__init__()
        self.heads, self.head_dim = heads, dim // heads
        self.to_qkv = SVDQW4A4Linear(dim, 3 * dim, **kw)
        self.norm_q = nn.RMSNorm(self.head_dim, eps=1e-6, dtype=kw["torch_dtype"], device=kw["device"])
        self.norm_k = nn.RMSNorm(self.head_dim, eps=1e-6, dtype=kw["torch_dtype"], device=kw["device"])
        # diffusers' FluxAttention keeps `to_out = [Linear, Dropout]` in joint blocks (checkpoint key `attn.to_out.0`); the
        # single blocks' output projection is the V2 key `attn.to_out` (transformer_flux_v2.py:564-625)
        self.to_out = nn.ModuleList([SVDQW4A4Linear(dim, dim, **kw), nn.Identity()]) if joint else SVDQW4A4Linear(dim, dim, **kw)
        self.joint = joint
        self.added_kv_proj_dim = dim if joint else None  # the attribute the reference's processors test (flux.py:84,177)
        if joint:
            self.add_qkv_proj = SVDQW4A4Linear(dim, 3 * dim, **kw)
            self.norm_added_q = nn.RMSNorm(self.head_dim, eps=1e-6, dtype=kw["torch_dtype"], device=kw["device"])
            self.norm_added_k = nn.RMSNorm(self.head_dim, eps=1e-6, dtype=kw["torch_dtype"], device=kw["device"])
            self.to_add_out = SVDQW4A4Linear(dim, dim, **kw)

    # "svdq": this library's attention kernel on the packed QKV (+ V^T side output of the QKV GEMM);
    # "sdpa": torch's scaled_dot_product_attention (the reference's "flashattn2" processor role,
    # models/attention_processors/flux.py:24-59).  "svdq" needs B == 1, head_dim 128, tokens % 128 == 0.
    attention_impl = "svdq"
    # True: the text and image stream's projections of a joint block share one GEMM launch each (svdq_gemm_args.wgt2)
    grouped = True  # plain class attribute (set False for A/B runs); nothing is read from the environment
    # True: the attention epilogue emits the output projection's quantised activation (svdq_attention_args.qact)
    fused_out_quant = True
    # True: the engine pads both token streams to 256 rows so that every token count runs the fused path (False: A/B -- token counts that
    # are not a multiple of 128 then take torch's SDPA)
    padded_tokens = True

    @property
    def out_proj(self) -> SVDQW4A4Linear:
        return self.to_out[0] if self.joint else self.to_out

    def _use_svdq(self, B, tokens):
        return self.attention_impl == "svdq" and B == 1 and self.head_dim == 128 and tokens % 128 == 0

    def forward(self, hidden, encoder_hidden=None, rotary=None, ln=None, ln_ctx=None, quantized=None, kv_valid=None):
        """``ln`` / ``ln_ctx`` = (stats, scale, shift): the inputs are the UN-normalised streams and the
        AdaLayerNormZero front end runs inside the QKV projections' quantiser.  ``kv_valid``: the real key rows when the
        streams are padded to 256 rows (the engine pads every token count onto this path; ``ops.attention.kv_valid_ranges``)."""
        B = hidden.shape[0]
        hd = self.heads * self.head_dim
        t_txt = encoder_hidden.shape[1] if self.joint else 0
        tokens = t_txt + hidden.shape[1]
        svdq = self._use_svdq(B, tokens)
        qs = q_prescale(self.head_dim) if svdq else 0.0  # the QKV GEMM emits Q times scale * log2(e): the attention kernel's fast geometry
        qkv = torch.empty(B, tokens, 3 * hd, dtype=hidden.dtype, device=hidden.device)
        vt = torch.empty(hd, tokens, dtype=hidden.dtype, device=hidden.device) if svdq else None
        grouped = False
        if self.joint:
            # both projections write straight into one [txt; img] buffer (B == 1): no torch.cat round trip
            rot_img, rot_txt = rotary[0], rotary[1]
            if self.grouped and B == 1 and len(rotary) > 2:  # one launch for both streams (rows: text, then image)
                grouped = fused_qkv_norm_rottary_pair(encoder_hidden, self.add_qkv_proj, self.norm_added_q, self.norm_added_k,
                                                      hidden, self.to_qkv, self.norm_q, self.norm_k, rotary[2], qkv[0],
                                                      out_vt=vt, ln_a=ln_ctx, ln_b=ln, q_scale=qs)
            if not grouped:
                fused_qkv_norm_rottary(hidden, self.to_qkv, self.norm_q, self.norm_k, rot_img, output=qkv[0, t_txt:],
                                       out_vt=vt[:, t_txt:] if svdq else None, ln=ln, q_scale=qs)
                fused_qkv_norm_rottary(encoder_hidden, self.add_qkv_proj, self.norm_added_q, self.norm_added_k, rot_txt,
                                       output=qkv[0, :t_txt], out_vt=vt[:, :t_txt] if svdq else None, ln=ln_ctx, q_scale=qs)
        else:
            fused_qkv_norm_rottary(hidden, self.to_qkv, self.norm_q, self.norm_k, rotary, output=qkv.view(B * tokens, -1),
                                   out_vt=vt, ln=ln, quantized=quantized, q_scale=qs)
        pool = None
        if svdq and self.fused_out_quant and B == 1 and (not self.joint or (self.grouped and _pair_compatible(self.to_add_out, self.out_proj))):
            src = ln_ctx if self.joint else ln  # the pool of the stream whose rows come first carries the scratch
            qpool = src[3] if src is not None and len(src) > 3 else None
            qres = attention_packed_quantized(qkv[0], vt, self.heads, self.out_proj, lin_first=self.to_add_out if self.joint else None,
                                              split_rows=t_txt, pool=qpool, q_prescaled=True, kv_valid=kv_valid)
            if qres is not None:  # the 16-bit attention output never exists: straight into the output projection(s)
                if self.joint:
                    ca, a = linear_pair_quantized(*qres, self.to_add_out, self.out_proj, t_txt)
                    return a, ca
                return self.out_proj.forward_quant(*qres).view(B, tokens, -1)
        if svdq:  # the same launch clears the low-rank accumulators of the output projections' quantisers
            zf = _pad256(hidden.shape[1]) * self.out_proj.rank + (_pad256(t_txt) * self.to_add_out.rank if self.joint else 0)
            o, pool = attention_packed(qkv[0], vt, self.heads, zero_floats=zf, q_prescaled=True, kv_valid=kv_valid)
            o = o.unsqueeze(0)
        else:
            q, k, v = qkv.chunk(3, dim=-1)
            shp = (B, -1, self.heads, self.head_dim)
            o = F.scaled_dot_product_attention(q.view(shp).transpose(1, 2), k.view(shp).transpose(1, 2),
                                               v.view(shp).transpose(1, 2), dropout_p=0.0, is_causal=False)
            o = o.transpose(1, 2).reshape(B, -1, hd)
        if self.joint:
            if self.grouped and B == 1:
                ca, a = linear_pair(o[:, :t_txt], self.to_add_out, o[:, t_txt:], self.out_proj, pool=pool)
                return a, ca
            return self.out_proj(o[:, t_txt:], pool=pool), self.to_add_out(o[:, :t_txt], pool=pool)
        return self.out_proj(o, pool=pool)


class _GELUProj(nn.Module):
    """``net.0`` of a diffusers FeedForward(activation_fn="gelu-approximate"): holds ``proj``; the activation itself runs in
    the projection's GEMM epilogue."""

    def __init__(self, dim, hidden, kw):
        super().__init__()
        self.proj = SVDQW4A4Linear(dim, hidden, **kw)


class _FeedForward(nn.Module):
    """fc1 -> GELU(tanh) -> fc2 with the requantisation fused into fc1's epilogue (reference: NunchakuFeedForward,
    models/attention.py:76-123).  Module names are diffusers' ``net = [GELU(proj), Dropout, Linear]``: checkpoint keys
    ``ff.net.0.proj.*`` / ``ff.net.2.*``."""

    def __init__(self, dim, kw):
        super().__init__()
        self.net = nn.ModuleList([_GELUProj(dim, 4 * dim, kw), nn.Identity(), SVDQW4A4Linear(4 * dim, dim, **{**kw, "act_unsigned": True})])

    @property
    def fc1(self) -> SVDQW4A4Linear:
        return self.net[0].proj

    @property
    def fc2(self) -> SVDQW4A4Linear:
        return self.net[2]

    def forward(self, x, ln=None):
        return fused_gelu_mlp(x, self.fc1, self.fc2, ln=ln)


class _AdaLNZero(nn.Module):
    """Parameter holder with the reference's / diffusers' module name: ``norm1.linear`` = the AdaLayerNormZero modulation
    projection, an AWQ W4A16 GEMV (normalization.py:85-98, linear.py:277-414).  The LayerNorm itself has no parameters and
    runs inside the quantiser (fused path) or as a torch op (block forward)."""

    def __init__(self, dim, chunks, dt, dev):
        super().__init__()
        self.linear = AWQW4A16Linear(dim, chunks * dim, torch_dtype=dt, device=dev)
        self.linear.out_chunks = chunks  # the GEMV writes the `chunks` [dim] vectors contiguously


class FluxJointBlockAMD(nn.Module):
    """reference: NunchakuFluxTransformerBlock (transformer_flux_v2.py:143-257); same sub-module names, so a V2 checkpoint's
    keys (``transformer_blocks.N.norm1.linear.qweight``, ``...attn.to_out.0.proj_up``, ``...ff.net.0.proj.wscales``) load as they are."""

    def __init__(self, dim, heads, kw):
        super().__init__()
        dt, dev = kw["torch_dtype"], kw["device"]
        self.norm1 = _AdaLNZero(dim, 6, dt, dev)
        self.norm1_context = _AdaLNZero(dim, 6, dt, dev)
        self.attn = FluxAttentionAMD(dim, heads, True, kw)
        self.ff = _FeedForward(dim, kw)
        self.ff_context = _FeedForward(dim, kw)
        self.dim = dim

    @property
    def mod(self) -> AWQW4A16Linear:
        return self.norm1.linear

    @property
    def mod_context(self) -> AWQW4A16Linear:
        return self.norm1_context.linear

    @staticmethod
    def _ln_mod(x, scale, shift):
        # NunchakuAdaLayerNormZero with scale_shift = 0 (normalization.py:85-98): the checkpoint's modulation bias
        # already carries the +1 of the scale
        return F.layer_norm(x, (x.shape[-1],), eps=1e-6) * scale[:, None] + shift[:, None]

    def forward(self, hidden, encoder_hidden, temb_act, rotary, stats=None, mods=None, kv_valid=None):
        """``mods`` = (mod, mod_context) outputs computed ahead of the block (one batched GEMV launch per step).
        ``stats`` = (image-stream, text-stream) LayerNorm statistics of the inputs: the fused path -- LayerNorm and
        modulation inside the quantisers, gated residual + next statistics in one element-wise pass (B == 1).
        Returns (encoder_hidden, hidden, stats)."""
        # normalization.py:85-98 -- emb.view(B, -1, 6).permute(2, 0, 1): interleaved chunks
        if stats is None:
            m = self.mod(temb_act).view(temb_act.shape[0], 6, -1).permute(1, 0, 2)
            c = self.mod_context(temb_act).view(temb_act.shape[0], 6, -1).permute(1, 0, 2)
            shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = m
            c_shift_msa, c_scale_msa, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = c
            n_h = self._ln_mod(hidden, scale_msa, shift_msa)
            n_e = self._ln_mod(encoder_hidden, c_scale_msa, c_shift_msa)
            a, ca = self.attn(n_h, n_e, rotary, kv_valid=kv_valid)
            hidden = hidden + gate_msa[:, None] * a  # transformer_flux_v2.py:230-251, op for op
            n_h = self._ln_mod(hidden, scale_mlp, shift_mlp)
            hidden = hidden + gate_mlp[:, None] * self.ff(n_h)
            encoder_hidden = encoder_hidden + c_gate_msa[:, None] * ca
            n_e = self._ln_mod(encoder_hidden, c_scale_mlp, c_shift_mlp)
            encoder_hidden = encoder_hidden + c_gate_mlp[:, None] * self.ff_context(n_e)
            if encoder_hidden.dtype == torch.float16:  # transformer_flux_v2.py: the fp16 joint block clips its text stream
                encoder_hidden = encoder_hidden.clip(-65504, 65504)
            return encoder_hidden, hidden, None
        (h_stats, h_pool), (e_stats, e_pool) = stats  # pools: fp32 zeros for the low-rank accumulators of the next calls
        m_out, c_out = mods if mods is not None else (self.mod(temb_act), self.mod_context(temb_act))
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = m_out.view(6, -1)
        c_shift_msa, c_scale_msa, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = c_out.view(6, -1)
        a, ca = self.attn(hidden, encoder_hidden, rotary, ln=(h_stats, scale_msa, shift_msa, h_pool),
                          ln_ctx=(e_stats, c_scale_msa, c_shift_msa, e_pool), kv_valid=kv_valid)
        mp_h, mp_e = _pad256(hidden.shape[1]), _pad256(encoder_hidden.shape[1])
        r_mlp = self.ff.fc1.rank + self.ff.fc2.rank          # fc1's quantiser + the GELU epilogue's accumulator for fc2
        r_mlp_c = self.ff_context.fc1.rank + self.ff_context.fc2.rank
        if self.attn.grouped and encoder_hidden.shape[1] % 256 == 0:
            # grouped launches: the text stream's pool carries the scratch of BOTH streams (its rows come first)
            encoder_hidden, e_stats, hidden, h_stats, e_pool = residual_gate_stats_pair(
                encoder_hidden, ca, c_gate_msa, hidden, a, gate_msa, zero_floats=(mp_e + mp_h) * r_mlp)
            ffc, ff = fused_gelu_mlp_pair(encoder_hidden, self.ff_context.fc1, self.ff_context.fc2, hidden, self.ff.fc1, self.ff.fc2,
                                          ln_a=(e_stats, c_scale_mlp, c_shift_mlp, e_pool), ln_b=(h_stats, scale_mlp, shift_mlp))
            encoder_hidden, e_stats, hidden, h_stats, e_pool = residual_gate_stats_pair(
                encoder_hidden, ffc, c_gate_mlp, hidden, ff, gate_mlp, zero_floats=(mp_e + mp_h) * (self.attn.to_qkv.rank + self.attn.out_proj.rank),
                clamp_fp16_a=True)  # the reference clips the text stream at the end of an fp16 joint block
            return encoder_hidden, hidden, ((h_stats, None), (e_stats, e_pool))
        hidden, h_stats, h_pool = residual_gate_stats(hidden, a, gate_msa, zero_floats=mp_h * r_mlp)
        hidden, h_stats, h_pool = residual_gate_stats(hidden, self.ff(hidden, ln=(h_stats, scale_mlp, shift_mlp, h_pool)), gate_mlp,
                                                      zero_floats=mp_h * self.attn.to_qkv.rank)  # next block's QKV quantiser
        encoder_hidden, e_stats, e_pool = residual_gate_stats(encoder_hidden, ca, c_gate_msa, zero_floats=mp_e * r_mlp_c)
        encoder_hidden, e_stats, e_pool = residual_gate_stats(
            encoder_hidden, self.ff_context(encoder_hidden, ln=(e_stats, c_scale_mlp, c_shift_mlp, e_pool)), c_gate_mlp,
            zero_floats=mp_e * self.attn.add_qkv_proj.rank, clamp_fp16=True)
        return encoder_hidden, hidden, ((h_stats, h_pool), (e_stats, e_pool))


class FluxSingleBlockAMD(nn.Module):
    """reference: NunchakuFluxSingleTransformerBlock (transformer_flux_v2.py:260-342): ``norm.linear``, ``attn.to_qkv``,
    ``attn.to_out``, ``mlp_fc1``, ``mlp_fc2``."""

    def __init__(self, dim, heads, kw):
        super().__init__()
        dt, dev = kw["torch_dtype"], kw["device"]
        self.norm = _AdaLNZero(dim, 3, dt, dev)  # AdaLayerNormZeroSingle.linear (normalization.py:155-165)
        self.mlp_fc1 = SVDQW4A4Linear(dim, 4 * dim, **kw)
        self.mlp_fc2 = SVDQW4A4Linear(4 * dim, dim, **{**kw, "act_unsigned": True})
        self.attn = FluxAttentionAMD(dim, heads, False, kw)

    @property
    def mod(self) -> AWQW4A16Linear:
        return self.norm.linear

    def forward(self, hidden, temb_act, rotary, stats=None, mods=None, kv_valid=None):
        if stats is None:
            shift, scale, gate = self.mod(temb_act).view(temb_act.shape[0], 3, -1).permute(1, 0, 2)
            n = F.layer_norm(hidden, (hidden.shape[-1],), eps=1e-6) * scale[:, None] + shift[:, None]
            mlp = fused_gelu_mlp(n, self.mlp_fc1, self.mlp_fc2)
            att = self.attn(n, rotary=rotary, kv_valid=kv_valid)
            out = hidden + gate[:, None] * (att + mlp)  # transformer_flux_v2.py:332-335
            if out.dtype == torch.float16:
                out = out.clip(-65504, 65504)
            return out, None
        shift, scale, gate = (mods if mods is not None else self.mod(temb_act)).view(3, -1)
        st, pool = stats
        ln = (st, scale, shift, pool)  # one LayerNorm + modulation, consumed by both projections' quantisers
        both = quantize_two(hidden, self.mlp_fc1, self.attn.to_qkv, ln=ln) if FluxAttentionAMD.grouped else None
        q_mlp, q_qkv = both if both is not None else (None, None)  # one quantiser launch for the two projections
        mlp = fused_gelu_mlp(hidden, self.mlp_fc1, self.mlp_fc2, ln=ln, quantized=q_mlp)
        att = self.attn(hidden, rotary=rotary, ln=ln, quantized=q_qkv, kv_valid=kv_valid)
        # hidden + gate * (att + mlp), the next block's statistics and its three low-rank accumulators, one pass
        hidden, st, pool = residual_gate_stats(hidden, att, gate, b=mlp, zero_floats=_pad256(hidden.shape[1]) * (
            self.mlp_fc1.rank + self.mlp_fc2.rank + self.attn.to_qkv.rank + self.attn.out_proj.rank), clamp_fp16=True)
        return hidden, (st, pool)


def _ids_versions(*ids):
    """version counters of the position-id tensors, or None when one of them does not track a version (inference tensors raise on ._version)"""
    try:
        return tuple(None if t is None else t._version for t in ids) if not any(t is not None and t.is_inference() for t in ids) else None
    except RuntimeError:
        return None


class FluxEngineMixin:
    """Everything of the FLUX.1 transformer that is not construction: the denoising-step forward over the module tree
    ``x_embedder / context_embedder / time_text_embed / transformer_blocks / single_transformer_blocks / norm_out / proj_out``
    (diffusers' names), the runtime-LoRA entry points and the synthetic initialiser.  Shared by the stand-alone
    :class:`FluxTransformerAMD` and -- when diffusers is importable -- the ``diffusers.FluxTransformer2DModel`` subclass of
    nunchaku_amd/models/transformer_flux.py, whose sub-modules are these same classes."""

    # True: AdaLayerNormZero runs inside the quantisers and the gated residuals are one fused pass each
    # (svdq_quantize_args.ln_stats, svdq_residual_gate_stats); False: the reference's torch-op sequence.
    fused_norm = True
    # True: all modulation GEMVs of a step in one batched launch before the first block
    batched_mods = True

    def _build_engine(self, num_layers=19, num_single_layers=38, dim=3072, heads=24, in_channels=64,
                      joint_attention_dim=4096, pooled_projection_dim=768, rank=32, guidance_embeds=True,
                      axes_dims_rope=(16, 56, 56), torch_dtype=torch.bfloat16, device="cuda"):
        """Create (or replace) the module tree on ``device``; parameters are uninitialised (load a checkpoint next)."""
        kw = dict(rank=rank, torch_dtype=torch_dtype, device=device)
        self.dim, self.axes = dim, tuple(axes_dims_rope)
        self.x_embedder = nn.Linear(in_channels, dim, dtype=torch_dtype, device=device)
        self.context_embedder = nn.Linear(joint_attention_dim, dim, dtype=torch_dtype, device=device)
        # module names = diffusers' FluxTransformer2DModel / the reference's V2 model: V2 checkpoints load key for key
        self.time_text_embed = _TimeTextEmbed(dim, pooled_projection_dim, guidance_embeds, torch_dtype, device)
        self.transformer_blocks = nn.ModuleList([FluxJointBlockAMD(dim, heads, kw) for _ in range(num_layers)])
        self.single_transformer_blocks = nn.ModuleList([FluxSingleBlockAMD(dim, heads, kw) for _ in range(num_single_layers)])
        self.norm_out = _AdaLNContinuous(dim, torch_dtype, device)
        self.proj_out = nn.Linear(dim, in_channels, dtype=torch_dtype, device=device)
        self.dtype_ = torch_dtype

    # short names used throughout this package (and by its tests / tools)
    @property
    def blocks(self):
        return self.transformer_blocks

    @property
    def single_blocks(self):
        return self.single_transformer_blocks

    @property
    def time_embed(self):
        return self.time_text_embed.timestep_embedder

    @property
    def guidance_embed(self):
        return self.time_text_embed.guidance_embedder

    @property
    def text_embed(self):
        return self.time_text_embed.text_embedder

    @property
    def norm_out_mod(self):
        return self.norm_out.linear

    def svdq_layers(self):
        return [m for m in self.modules() if isinstance(m, SVDQW4A4Linear)]

    def set_attention_impl(self, impl: str, attn_func=None):
        """reference: NunchakuFluxTransformer2dModel.set_attention_impl (transformer_flux.py:648-667).  ``"nunchaku-fp16"``:
        the attention kernel of this library fed by the QKV epilogue (its role on MI355X; bf16 and fp16);
        ``"flashattn2"``: ``torch.nn.functional.scaled_dot_product_attention``."""
        table = {"nunchaku-fp16": "svdq", "svdq": "svdq", "flashattn2": "sdpa", "sdpa": "sdpa"}
        if impl == "custom" or attn_func is not None:
            raise NotImplementedError("set_attention_impl: custom attention functions are not supported")
        if impl not in table:
            raise ValueError(f"set_attention_impl: unknown implementation {impl!r}")
        for m in self.modules():
            if isinstance(m, FluxAttentionAMD):
                m.attention_impl = table[impl]

    # runtime LoRA (reference: NunchakuFluxTransformer2dModel.update_lora_params / set_lora_strength,
    # transformer_flux.py:783-855): per-layer factors in logical layout widen the low-rank branch of that layer
    def update_lora_params(self, lora: dict, strength: float = 1.0):
        """``lora``: module name (e.g. ``"blocks.0.attn.to_qkv"``) -> ``(down [r, in], up [out, r])``."""
        mods = dict(self.named_modules())
        self.reset_lora()
        for name, (down, up) in lora.items():
            if not isinstance(mods.get(name), SVDQW4A4Linear):
                raise KeyError(f"update_lora_params: {name} is not an SVDQW4A4Linear of this model")
            mods[name].set_lora(down, up, strength)

    def set_lora_strength(self, strength: float):
        for m in self.svdq_layers():
            if m._base_lowrank is not None:
                m.set_lora_strength(strength)

    def reset_lora(self):
        for m in self.svdq_layers():
            m.reset_lora()

    @torch.no_grad()
    def init_synthetic_(self, seed: int = 0, repack: bool = True, codes: str = "uniform"):
        """Random-init weights of FLUX shape (no checkpoints in this environment): int4 codes uniform or with the
        distribution of a quantised Gaussian residual (``codes``: models/linear.py ``synthetic_codes_``),
        scales/low-rank factors small so activations stay O(1).  Parameters are written in the
        checkpoint layout (random nibbles are random int4 codes) and repacked like a real checkpoint."""
        dev = self.proj_out.weight.device
        g = torch.Generator(device=dev).manual_seed(seed)

        def rnd(shape, scale):
            return torch.randn(shape, generator=g, device=dev) * scale

        def uni(shape):
            return torch.rand(shape, generator=g, device=dev)

        for m in self.modules():
            if isinstance(m, SVDQW4A4Linear):
                K = m.in_features
                synthetic_codes_(m.qweight, m.wscales, K, g, codes)  # scaled so that |W row| ~ 1/sqrt(K)
                if m.bias is not None:
                    m.bias.copy_(rnd(m.bias.shape, 0.02))
                m.smooth_factor.copy_(uni((K,)) + 0.5)
                m.smooth_factor_orig.copy_(m.smooth_factor)
                m.proj_down.copy_(rnd(m.proj_down.shape, 0.5 / math.sqrt(K)))
                m.proj_up.copy_(rnd(m.proj_up.shape, 0.5 / math.sqrt(m.rank)))
                m._amd_layout = False
                if repack:  # False: stay in the checkpoint layout (repacked lazily on first use, like a loaded checkpoint)
                    m.repack_()
            elif isinstance(m, AWQW4A16Linear):
                # uniform 4-bit codes (std 4.6) centred by the zero point: weights ~ 1/sqrt(K)
                sc = 1.0 / (4.6 * math.sqrt(m.in_features))
                m.qweight.copy_(torch.randint(-2 ** 31, 2 ** 31, m.qweight.shape, generator=g, device=dev, dtype=torch.int64))
                m.wscales.copy_((uni(m.wscales.shape) * 0.5 + 0.75) * sc)
                m.wzeros.copy_(m.wscales.float() * -7.5)
                # nunchaku checkpoints carry the +1 of every modulation scale in the bias (scale_shift = 0,
                # normalization.py:24-25): chunks (shift, SCALE, gate[, shift, SCALE, gate]) are interleaved per channel
                m.bias.zero_()
                chunks = m.out_features // m.in_features
                m.bias.view(-1, chunks)[:, 1::3] = 1.0  # checkpoint (interleaved) order; out_chunks only permutes the OUTPUT
            elif isinstance(m, nn.Linear):
                m.weight.copy_(rnd(m.weight.shape, 1.0 / math.sqrt(m.in_features)))
                m.bias.zero_()
            elif isinstance(m, nn.RMSNorm):
                m.weight.fill_(1.0)
        return self

    def engine_forward(self, hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids,
                       guidance=None, controlnet_block_samples=None, controlnet_single_block_samples=None, controlnet_blocks_repeat=False):
        """hidden_states [1, T_img, 64]; encoder_hidden_states [1, T_txt, 4096]; pooled [1, 768];
        timestep/guidance [1]; img_ids [T_img, 3]; txt_ids [T_txt, 3]  ->  [1, T_img, 64]
        (transformer_flux_v2.py:430-561; batch 1 -- the fused QKV epilogue takes one rotary table).
        ``controlnet_block_samples`` / ``controlnet_single_block_samples``: lists of ``[1, T_img, dim]`` residuals added to the image
        stream behind the joint / single blocks with diffusers' indexing (``FluxTransformer2DModel.forward``: sample
        ``i // ceil(blocks / samples)``, or ``i % samples`` with ``controlnet_blocks_repeat``)."""
        dt = self.dtype_
        if hidden_states.shape[0] > 1:
            # The fused QKV epilogue takes ONE rotary table and the operand buffers of a launch belong to one sample
            # (reference: rotary_emb.shape[0] * shape[1] == M assert, launch_impl.cuh:353): a batch
            # is a loop over samples here -- the data-parallel unit of this library is the replica, not the batch axis.
            def per(t, i):
                return t[i:i + 1] if t is not None and t.dim() > 0 and t.shape[0] == hidden_states.shape[0] else t
            def per_list(ts, i):
                return None if ts is None else [per(t, i) for t in ts]
            return torch.cat([self.engine_forward(hidden_states[i:i + 1], encoder_hidden_states[i:i + 1], pooled_projections[i:i + 1],
                                           per(timestep, i), img_ids, txt_ids, per(guidance, i), per_list(controlnet_block_samples, i),
                                           per_list(controlnet_single_block_samples, i), controlnet_blocks_repeat)
                              for i in range(hidden_states.shape[0])], dim=0)
        hidden = self.x_embedder(hidden_states)
        # diffusers casts timestep / guidance to the model dtype BEFORE the x1000 (transformer_flux.py: timestep.to(dtype) * 1000)
        temb = self.time_embed(timestep_embedding(timestep.to(dt) * 1000).to(dt))
        if self.guidance_embed is not None:
            temb = temb + self.guidance_embed(timestep_embedding(guidance.to(dt) * 1000).to(dt))
        temb = temb + self.text_embed(pooled_projections)
        temb_act = F.silu(temb)
        enc = self.context_embedder(encoder_hidden_states)

        t_txt, t_img = enc.shape[1], hidden.shape[1]
        attn0 = (self.blocks[0] if len(self.blocks) else self.single_blocks[0]).attn
        pad_streams = FluxAttentionAMD.padded_tokens and attn0.attention_impl == "svdq" and attn0.head_dim == 128
        # the rotary tables depend on the position ids only: a denoise loop passes the same two tensor OBJECTS every step -- built once and kept
        # (one entry; the cache holds the id tensors themselves, so "the same object, unmodified" cannot be a recycled address).  Not cached:
        #  * ids whose version counter cannot be read (tensors created under torch.inference_mode() do not track one): "unmodified" is unknowable;
        #  * while the stream is capturing: the tables are computed INSIDE the graph (its pool owns them and ids passed as static graph inputs
        #    keep their meaning on replay); a graph must never bake in pointers to tables that only this one-entry cache keeps alive -- a later
        #    eager call with other ids would free them under the graph (same hazard as _Workspace.captured, _C.release_workspaces).
        vers = _ids_versions(txt_ids, img_ids)
        use_cache = vers is not None and not (hidden.is_cuda and torch.cuda.is_current_stream_capturing())
        key = (vers, pad_streams)
        cached = getattr(self, "_rot_cache", None) if use_cache else None
        if cached is not None and cached[0] is txt_ids and cached[1] is img_ids and cached[2] == key:
            rot_txt, rot_img, rot_all, p_txt, p_img = cached[3]
        else:
            rot = flux_pos_embed(torch.cat([txt_ids, img_ids], dim=0), self.axes)  # [1, T, 64, 1, 2]
            rot_t, rot_i = pad_tensor(rot[:, :t_txt], 256, 1), pad_tensor(rot[:, t_txt:], 256, 1)
            rot_txt, rot_img = pack_rotemb(rot_t), pack_rotemb(rot_i)
            p_txt, p_img = rot_t.shape[1], rot_i.shape[1]
            # joint table: every stream on a 256-row boundary when the streams are padded (below), the plain concatenation otherwise
            rot_all = pack_rotemb(torch.cat([rot_t, rot_i], dim=1)) if pad_streams and kv_valid_ranges(t_txt, t_img) is not None \
                else pack_rotemb(pad_tensor(rot, 256, 1))
            if use_cache:
                self._rot_cache = (txt_ids, img_ids, key, (rot_txt, rot_img, rot_all, p_txt, p_img))
        # EVERY token count runs the hot path (the reference pads any M to 256 rows, Linear.cpp:445-446, and masks the padded K rows of its
        # attention, epilogues.cuh:427-550): both streams are padded to 256 rows with zero tokens right behind the embedders -- the joint
        # sequence is [text | pad | image | pad], every stream starts on a 256-row boundary as the grouped launches need -- every launch of
        # the step sees the shapes the parity suite and the bench exercise, the attention kernel masks the padded keys (kv_valid), and the
        # real image rows are sliced out at the end.  A padded row is a token nobody attends to: it stays finite and touches no real row.
        kv_valid = kv_valid_ranges(t_txt, t_img) if pad_streams else None
        if kv_valid is not None:
            enc, hidden = F.pad(enc, (0, 0, 0, p_txt - t_txt)), F.pad(hidden, (0, 0, 0, p_img - t_img))
        else:
            p_txt = t_txt

        fused = self.fused_norm and hidden.shape[0] == 1
        stats = ((residual_gate_stats(hidden)[1], None), (residual_gate_stats(enc)[1], None)) if fused else None
        # every modulation projection depends on the timestep embedding only: one batched GEMV launch for the whole step
        mods = awq_gemv_w4a16_batched(temb_act, [m for b in self.blocks for m in (b.mod, b.mod_context)] +
                                      [b.mod for b in self.single_blocks]) if fused and self.batched_mods else None
        nj = len(self.blocks)

        def control(samples, i, n_blocks):
            """diffusers' choice of the ControlNet residual behind block i, as a [1, rows of the (padded) image stream, dim] tensor"""
            n = len(samples)
            smp = samples[i % n] if controlnet_blocks_repeat else samples[i // -(-n_blocks // n)]
            return F.pad(smp.to(hidden.dtype), (0, 0, 0, hidden.shape[1] - smp.shape[1])) if smp.shape[1] != hidden.shape[1] else smp.to(hidden.dtype)

        for i, blk in enumerate(self.blocks):
            enc, hidden, stats = blk(hidden, enc, temb_act, (rot_img, rot_txt, rot_all), stats,
                                     mods=(mods[2 * i], mods[2 * i + 1]) if mods is not None else None, kv_valid=kv_valid)
            if controlnet_block_samples is not None:
                # hidden_states + sample (one 16-bit add); the fused path needs the LayerNorm statistics of the sum: the same pass
                hidden, h_stats = residual_gate_stats(hidden, control(controlnet_block_samples, i, nj), want_stats=fused)
                if fused:
                    stats = ((h_stats, stats[0][1]), stats[1])
        t_pad = enc.shape[1]
        hidden = torch.cat([enc, hidden], dim=1)
        stats = (torch.cat([stats[1][0], stats[0][0]], dim=0), None) if fused else None  # [txt; img] row order
        for i, blk in enumerate(self.single_blocks):
            hidden, stats = blk(hidden, temb_act, rot_all, stats, mods=mods[2 * nj + i] if mods is not None else None, kv_valid=kv_valid)
            if controlnet_single_block_samples is not None:
                img_rows = hidden[:, t_pad:]
                smp = control(controlnet_single_block_samples, i, len(self.single_blocks))
                _, i_stats = residual_gate_stats(img_rows, F.pad(smp, (0, 0, 0, img_rows.shape[1] - smp.shape[1])), want_stats=fused)
                if fused:
                    stats[0][t_pad:] = i_stats
        hidden = hidden[:, p_txt:p_txt + t_img]
        scale, shift = self.norm_out_mod(temb_act).chunk(2, dim=-1)  # AdaLayerNormContinuous
        hidden = F.layer_norm(hidden, (self.dim,), eps=1e-6) * (1 + scale[:, None]) + shift[:, None]
        return self.proj_out(hidden)


class FluxTransformerAMD(nn.Module, FluxEngineMixin):
    """One denoising step: ``forward(latents, text states, pooled text, timestep, guidance, ids)`` (stand-alone; no diffusers)."""

    def __init__(self, num_layers=19, num_single_layers=38, dim=3072, heads=24, in_channels=64,
                 joint_attention_dim=4096, pooled_projection_dim=768, rank=32, guidance_embeds=True,
                 axes_dims_rope=(16, 56, 56), torch_dtype=torch.bfloat16, device="cuda"):
        super().__init__()
        self._build_engine(num_layers, num_single_layers, dim, heads, in_channels, joint_attention_dim, pooled_projection_dim,
                           rank, guidance_embeds, axes_dims_rope, torch_dtype, device)

    def forward(self, hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids, guidance=None):
        return self.engine_forward(hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids, guidance)
