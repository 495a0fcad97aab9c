"""Checkpoint plumbing for the FLUX transformer.

Reference: ``NunchakuFluxTransformer2DModelV2.from_pretrained`` (nunchaku/models/transformers/transformer_flux_v2.py:
373-428) = safetensors file with a ``config`` / ``quantization_config`` metadata header -> ``convert_flux_state_dict``
(:564-625, legacy C++-model key names -> V2 module names) -> ``patch_scale_key`` (transformers/utils.py:151-173) ->
``load_state_dict``.  Here the same file loads into :class:`FluxTransformerAMD`: the key conversion goes from either
naming to V2 naming -- which is this package's own (module names follow diffusers' FluxTransformer2DModel) --, NVFP4-only tensors (``wtscale`` / ``wcscales``) are dropped, and
the SVDQuant tensors are re-laid out for CDNA4 lazily by ``SVDQW4A4Linear.repack_()`` on first use (AWQ tensors are
consumed as stored).  Pure host-side code: no kernels, testable without a GPU.
"""

from __future__ import annotations

import json
import os
import re

import torch

from .flux import FluxTransformerAMD

# (legacy name, V2 name) of the sub-modules of a joint block / a single block.  FluxTransformerAMD's own module names ARE the V2
# (diffusers) names, so a V2 state dict loads key for key and only the legacy (C++ model) spelling needs conversion.
_JOINT = [
    ("norm1.linear", "norm1.linear"),
    ("norm1_context.linear", "norm1_context.linear"),
    ("qkv_proj_context", "attn.add_qkv_proj"),
    ("qkv_proj", "attn.to_qkv"),
    ("norm_added_q", "attn.norm_added_q"),
    ("norm_added_k", "attn.norm_added_k"),
    ("norm_q", "attn.norm_q"),
    ("norm_k", "attn.norm_k"),
    ("out_proj_context", "attn.to_add_out"),
    ("out_proj", "attn.to_out.0"),
    ("mlp_context_fc1", "ff_context.net.0.proj"),
    ("mlp_context_fc2", "ff_context.net.2"),
    ("mlp_fc1", "ff.net.0.proj"),
    ("mlp_fc2", "ff.net.2"),
]
_SINGLE = [
    ("norm.linear", "norm.linear"),
    ("qkv_proj", "attn.to_qkv"),
    ("norm_q", "attn.norm_q"),
    ("norm_k", "attn.norm_k"),
    ("out_proj", "attn.to_out"),
    ("mlp_fc1", "mlp_fc1"),
    ("mlp_fc2", "mlp_fc2"),
]
_TOP = ("time_text_embed.timestep_embedder", "time_text_embed.guidance_embedder", "time_text_embed.text_embedder",
        "norm_out.linear", "x_embedder", "context_embedder", "proj_out")  # the 16-bit parts: diffusers names, unchanged
# legacy -> V2 parameter names of an SVDQuant layer (transformer_flux_v2.py:595-601)
_PARAM = [("lora_down", "proj_down"), ("lora_up", "proj_up"), ("smooth_orig", "smooth_factor_orig"), ("smooth", "smooth_factor")]
_DROP = ("wtscale", "wcscales")  # NVFP4 only (patch_scale_key pops / defaults them)


def _convert_param(rest: str) -> str:
    for old, new in _PARAM:
        if rest == old:
            return new
    return rest


def convert_key(key: str) -> str | None:
    """One checkpoint key (legacy or V2 naming) -> the V2 / FluxTransformerAMD state-dict key, or None for a tensor this
    model does not use.  Raises ``KeyError`` for a name it does not recognise."""
    if key.rsplit(".", 1)[-1] in _DROP:
        return None
    m = re.match(r"(single_transformer_blocks|transformer_blocks)\.(\d+)\.(.+)$", key)
    if m:
        kind, idx, rest = m.groups()
        table = _SINGLE if kind.startswith("single") else _JOINT
        for legacy, v2 in table:
            for src in (v2, legacy):
                if rest.startswith(src + "."):
                    return f"{kind}.{idx}.{v2}.{_convert_param(rest[len(src) + 1:])}"
        raise KeyError(f"unrecognised FLUX block tensor: {key}")
    for src in _TOP:
        if key.startswith(src + "."):
            return key
    raise KeyError(f"unrecognised FLUX tensor: {key}")


def convert_flux_state_dict(state_dict: dict) -> dict:
    """Checkpoint state dict (legacy ``NunchakuFluxTransformer2dModel`` or V2 naming) -> V2 naming (= this package's)."""
    out = {}
    for k, v in state_dict.items():
        nk = convert_key(k)
        if nk is None:
            continue
        if nk in out:
            raise KeyError(f"two checkpoint tensors map to {nk}")
        out[nk] = v
    return out


def export_legacy_state_dict(model: FluxTransformerAMD) -> dict:
    """Inverse of :func:`convert_flux_state_dict`: V2 names -> the reference's legacy names.  Works before and after the first
    forward: ``state_dict()`` of a repacked ``SVDQW4A4Linear`` returns checkpoint-layout tensors."""
    inv_param = {new: old for old, new in _PARAM}
    out = {}
    for k, v in model.state_dict().items():
        m = re.match(r"(single_transformer_blocks|transformer_blocks)\.(\d+)\.(.+)$", k)
        if m:
            kind, idx, rest = m.groups()
            for legacy, v2 in (_SINGLE if kind.startswith("single") else _JOINT):
                if rest.startswith(v2 + "."):
                    p = rest[len(v2) + 1:]
                    out[f"{kind}.{idx}.{legacy}.{inv_param.get(p, p)}"] = v
                    break
            else:
                raise KeyError(k)
            continue
        if not k.startswith(tuple(t + "." for t in _TOP)):
            raise KeyError(k)
        out[k] = v
    return out


def model_from_config(config: dict, rank: int = 32, torch_dtype: torch.dtype = torch.bfloat16, device="cuda") -> FluxTransformerAMD:
    """diffusers ``FluxTransformer2DModel`` config (the ``config`` metadata entry of a nunchaku safetensors file)."""
    heads, hd = config.get("num_attention_heads", 24), config.get("attention_head_dim", 128)
    return FluxTransformerAMD(
        num_layers=config.get("num_layers", 19), num_single_layers=config.get("num_single_layers", 38), dim=heads * hd,
        heads=heads, in_channels=config.get("in_channels", 64), joint_attention_dim=config.get("joint_attention_dim", 4096),
        pooled_projection_dim=config.get("pooled_projection_dim", 768), rank=rank,
        guidance_embeds=config.get("guidance_embeds", True), axes_dims_rope=tuple(config.get("axes_dims_rope", (16, 56, 56))),
        torch_dtype=torch_dtype, device=device)


def load_flux_state_dict(model: FluxTransformerAMD, state_dict: dict, strict: bool = True) -> FluxTransformerAMD:
    """Convert + ``load_state_dict``; dtype mismatches are errors as in ``patch_scale_key`` (utils.py:165-167)."""
    sd = convert_flux_state_dict(state_dict)
    own = dict(model.named_parameters())  # (the live tensors; state_dict() would convert repacked layers back first)
    own.update(dict(model.named_buffers()))
    for k, v in sd.items():
        if k in own and own[k].dtype != v.dtype:
            raise TypeError(f"{k}: checkpoint dtype {v.dtype} != model dtype {own[k].dtype}")
    model.load_state_dict(sd, strict=strict)
    return model


def from_pretrained(path: str | os.PathLike, device="cuda", torch_dtype: torch.dtype = torch.bfloat16) -> FluxTransformerAMD:
    """Load a nunchaku FLUX ``.safetensors`` checkpoint (int4; ``svdq-int4_r32-flux.1-*.safetensors``)."""
    from safetensors import safe_open

    path = os.fspath(path)
    if not path.endswith((".safetensors", ".sft")):
        raise ValueError("only safetensors checkpoints are supported (transformer_flux_v2.py:405-407)")
    sd = {}
    with safe_open(path, framework="pt", device="cpu") as f:
        meta = f.metadata() or {}
        for k in f.keys():
            sd[k] = f.get_tensor(k)
    config = json.loads(meta.get("config", "{}"))
    qcfg = json.loads(meta.get("quantization_config", "{}"))
    if any(k.endswith(".wcscales") for k in sd) or qcfg.get("weight", {}).get("dtype", "int4") not in ("int4",):
        raise NotImplementedError("NVFP4 checkpoints need Blackwell's block-scaled mma; use the int4 checkpoint on MI355X")
    model = model_from_config(config, rank=qcfg.get("rank", 32), torch_dtype=torch_dtype, device=device)
    return load_flux_state_dict(model, sd)
