"""diffusers-facing surface of the FLUX transformer.

Reference: ``NunchakuFluxTransformer2DModelV2`` (nunchaku/models/transformers/transformer_flux_v2.py:345-561), a
``diffusers.FluxTransformer2DModel`` subclass + ``NunchakuModelLoaderMixin`` (transformers/utils.py:26-59) that a
``FluxPipeline`` takes as its ``transformer``:

    transformer = NunchakuFluxTransformer2DModelV2.from_pretrained("svdq-int4_r32-flux.1-dev.safetensors")
    pipe = FluxPipeline.from_pretrained("black-forest-labs/FLUX.1-dev", transformer=transformer, torch_dtype=torch.bfloat16)

Two builds of the same class, chosen at import time:

* **diffusers importable**: a REAL ``FluxTransformer2DModel`` subclass.  ``from_pretrained`` follows the reference's flow --
  ``_build_model`` makes the diffusers skeleton from the checkpoint's ``config`` on the meta device (no 24 GB of bf16
  weights are ever allocated), ``_patch_model`` replaces its sub-modules by this package's (same module NAMES:
  ``transformer_blocks.N.attn.to_out.0``, ``...ff.net.0.proj``, ``time_text_embed.timestep_embedder`` ..., so PEFT / LoRA
  loaders keyed on diffusers names, ``ModelMixin.to()`` / ``dtype`` / ``device``, ``FrozenDict`` config, ``cache_context``
  and ``enable_model_cpu_offload`` see an ordinary diffusers model), then the converted state dict is loaded.
* **diffusers absent** (this container): the same class on ``torch.nn.Module`` -- keyword ``forward`` with the pipeline's
  argument names, a ``config`` namespace with the fields the pipeline reads, ``dtype`` / ``device`` and an output object with
  ``.sample``: what ``FluxPipeline`` duck-types.

Both share :class:`~nunchaku_amd.models.flux.FluxEngineMixin` (the forward over the module tree) and
:class:`NunchakuModelLoaderMixin`.  The legacy class name ``NunchakuFluxTransformer2dModel`` is an alias; its LoRA entry points
``update_lora_params`` / ``set_lora_strength`` live on the engine mixin.
"""

from __future__ import annotations

import json
import os
from types import SimpleNamespace

import torch
from torch import nn

from . import loader
from .flux import FluxEngineMixin

try:  # the guarded import the reference does unconditionally (transformer_flux_v2.py:12-24)
    from diffusers import FluxTransformer2DModel as _DiffusersFlux
    from diffusers.models.modeling_outputs import Transformer2DModelOutput

    HAVE_DIFFUSERS = True
except Exception:  # ImportError, or a diffusers build that cannot import on this platform
    _DiffusersFlux = None
    HAVE_DIFFUSERS = False

    class Transformer2DModelOutput:
        """Stand-in for ``diffusers.models.modeling_outputs.Transformer2DModelOutput`` (one field)."""

        def __init__(self, sample: torch.Tensor):
            self.sample = sample

        def __getitem__(self, i):
            return (self.sample,)[i]


_DEFAULT_CONFIG = dict(num_layers=19, num_single_layers=38, num_attention_heads=24, attention_head_dim=128, in_channels=64,
                       joint_attention_dim=4096, pooled_projection_dim=768, guidance_embeds=True, axes_dims_rope=(16, 56, 56),
                       patch_size=1, out_channels=None)


class NunchakuModelLoaderMixin:
    """reference: transformers/utils.py:26-59 -- ``_build_model(path) -> (transformer skeleton, state dict, metadata)`` from a
    nunchaku ``.safetensors`` file whose metadata carries the diffusers ``config`` and the ``quantization_config``."""

    @staticmethod
    def _read_safetensors(pretrained_model_name_or_path: str | os.PathLike):
        """Local file, or ``org/repo/file.safetensors`` on the Hugging Face hub (``hf_hub_download``; needs network access)."""
        from safetensors import safe_open

        path = os.fspath(pretrained_model_name_or_path)
        if not path.endswith((".safetensors", ".sft")):
            raise ValueError("Only safetensors are supported")  # transformer_flux_v2.py:405-407
        if not os.path.isfile(path):
            parts = path.split("/")
            if len(parts) < 3:
                raise FileNotFoundError(path)
            from huggingface_hub import hf_hub_download

            path = hf_hub_download(repo_id="/".join(parts[:2]), filename="/".join(parts[2:]))
        sd = {}
        with safe_open(path, framework="pt", device="cpu") as f:
            meta = f.metadata() or {}
            for k in f.keys():
                sd[k] = f.get_tensor(k)
        return sd, meta

    @classmethod
    def _build_model(cls, pretrained_model_name_or_path: str | os.PathLike, **kwargs):
        sd, meta = cls._read_safetensors(pretrained_model_name_or_path)
        config = json.loads(meta.get("config", "{}"))
        if HAVE_DIFFUSERS and _DiffusersFlux is not None and issubclass(cls, _DiffusersFlux):
            with torch.device("meta"):  # the skeleton only: _patch_model replaces every parametrised sub-module
                transformer = cls.from_config(config).to(kwargs.get("torch_dtype", torch.bfloat16))
        else:
            transformer = cls.__new__(cls)
            nn.Module.__init__(transformer)
            cfg = dict(_DEFAULT_CONFIG)
            cfg.update(config)
            transformer.config = SimpleNamespace(**cfg)  # FluxPipeline reads .in_channels and .guidance_embeds
        return transformer, sd, meta


_Base = _DiffusersFlux if HAVE_DIFFUSERS else nn.Module
_UNSET = object()  # "this configuration key was not passed by keyword"


class NunchakuFluxTransformer2DModelV2(_Base, FluxEngineMixin, NunchakuModelLoaderMixin):
    """``FluxPipeline``'s ``transformer`` on the MI355X SVDQuant path.  ``NunchakuFluxTransformer2DModelV2(config_dict)``
    builds an uninitialised model (synthetic weights: ``init_synthetic_``); ``from_pretrained`` loads a checkpoint."""

    def __init__(self, config: dict | None = None, rank: int = 32, torch_dtype: torch.dtype = torch.bfloat16, device="cuda", *,
                 patch_size=_UNSET, in_channels=_UNSET, out_channels=_UNSET, num_layers=_UNSET, num_single_layers=_UNSET,
                 attention_head_dim=_UNSET, num_attention_heads=_UNSET, joint_attention_dim=_UNSET, pooled_projection_dim=_UNSET,
                 guidance_embeds=_UNSET, axes_dims_rope=_UNSET, **diffusers_kwargs):
        """``NunchakuFluxTransformer2DModelV2(config_dict, device=...)`` builds a complete (uninitialised) model.  The diffusers
        configuration keys are ALSO named parameters: ``ConfigMixin.from_config`` -> ``extract_init_dict`` passes only the keys it finds in
        ``__init__``'s signature, so a checkpoint's ``num_layers`` / ``guidance_embeds`` ... reach the skeleton that ``_build_model`` makes
        on the meta device (with a bare ``**kwargs`` they were dropped and a default-size model was built).  Called that way --
        configuration by keyword, no ``config`` dict -- the constructor only makes the skeleton; ``from_pretrained`` patches it."""
        named = dict(patch_size=patch_size, in_channels=in_channels, out_channels=out_channels, num_layers=num_layers,
                     num_single_layers=num_single_layers, attention_head_dim=attention_head_dim, num_attention_heads=num_attention_heads,
                     joint_attention_dim=joint_attention_dim, pooled_projection_dim=pooled_projection_dim, guidance_embeds=guidance_embeds,
                     axes_dims_rope=axes_dims_rope)
        by_keyword = {k: v for k, v in named.items() if v is not _UNSET}
        by_keyword.update(diffusers_kwargs)
        cfg = dict(_DEFAULT_CONFIG)
        cfg.update(config or {})
        cfg.update(by_keyword)
        if HAVE_DIFFUSERS:
            # diffusers' own constructor builds the full-size bf16 module tree: only ever on the meta device (from_config under
            # torch.device("meta") in _build_model, or here); _patch_model then materialises this package's modules
            with torch.device("meta"):
                super().__init__(**{k: v for k, v in cfg.items() if v is not None or k == "out_channels"})
        else:
            nn.Module.__init__(self)
            self.config = SimpleNamespace(**cfg)
        if config is not None or not HAVE_DIFFUSERS or not by_keyword:
            self._patch_model(rank=rank, torch_dtype=torch_dtype, device=device)

    def _patch_model(self, rank: int = 32, torch_dtype: torch.dtype = torch.bfloat16, device="cuda", **kwargs):
        """reference :350-371 -- replace the blocks (and, here, the small 16-bit modules: the skeleton lives on the meta
        device) by this package's modules under the SAME names."""
        if kwargs.get("precision", "int4") not in ("int4",):
            raise NotImplementedError("NVFP4 checkpoints need Blackwell's block-scaled mma; use the int4 checkpoint on MI355X")
        c = self.config
        get = (lambda k: c[k]) if isinstance(c, dict) or hasattr(c, "keys") else (lambda k: getattr(c, k))
        heads, hd = get("num_attention_heads"), get("attention_head_dim")
        self._build_engine(num_layers=get("num_layers"), num_single_layers=get("num_single_layers"), dim=heads * hd, heads=heads,
                           in_channels=get("in_channels"), joint_attention_dim=get("joint_attention_dim"),
                           pooled_projection_dim=get("pooled_projection_dim"), rank=rank, guidance_embeds=get("guidance_embeds"),
                           axes_dims_rope=tuple(get("axes_dims_rope")), torch_dtype=torch_dtype, device=device)
        if HAVE_DIFFUSERS and hasattr(self, "pos_embed"):
            self.pos_embed = nn.Identity()  # the rotary tables are computed by the engine (embeddings.flux_pos_embed)
        return self

    if not HAVE_DIFFUSERS:  # ModelMixin provides these

        @property
        def dtype(self) -> torch.dtype:
            return self.dtype_

        @property
        def device(self) -> torch.device:
            return self.proj_out.weight.device

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str | os.PathLike, **kwargs):
        """A nunchaku ``.safetensors`` file, local or ``org/repo/file`` on the hub (reference :373-428)."""
        if kwargs.get("offload", False):
            raise NotImplementedError("Offload is not supported for FluxTransformer2DModelV2")  # as the reference, :398-399
        transformer, sd, meta = cls._build_model(pretrained_model_name_or_path, **kwargs)
        qcfg = json.loads(meta.get("quantization_config", "{}"))
        if any(k.endswith(".wcscales") for k in sd) or qcfg.get("weight", {}).get("dtype", "int4") not in ("int4",):
            raise NotImplementedError("NVFP4 checkpoints need Blackwell's block-scaled mma; use the int4 checkpoint on MI355X")
        transformer._patch_model(rank=qcfg.get("rank", 32), torch_dtype=kwargs.get("torch_dtype", torch.bfloat16),
                                 device=kwargs.get("device", "cuda"))
        return loader.load_flux_state_dict(transformer, sd)

    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor = None,
                pooled_projections: torch.Tensor = None, timestep: torch.Tensor = None, img_ids: torch.Tensor = None,
                txt_ids: torch.Tensor = None, guidance: torch.Tensor = None, joint_attention_kwargs=None,
                controlnet_block_samples=None, controlnet_single_block_samples=None, return_dict: bool = True,
                controlnet_blocks_repeat: bool = False):
        """The ``FluxPipeline`` call (reference :430-561): ``timestep`` arrives divided by 1000, ids as [T, 3] (a leading
        batch axis, deprecated in diffusers, is dropped as the reference does :505-517)."""
        if txt_ids is not None and txt_ids.ndim == 3:
            txt_ids = txt_ids[0]
        if img_ids is not None and img_ids.ndim == 3:
            img_ids = img_ids[0]
        if self.guidance_embed is not None and guidance is None:
            raise ValueError("this checkpoint has guidance embeddings: pass guidance")
        # ControlNet residuals: per-block additions to the image stream with diffusers' indexing (the reference's V2 forward raises here,
        # transformer_flux_v2.py:537-552; its legacy model and diffusers' FluxTransformer2DModel accept them)
        out = self.engine_forward(hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids, guidance,
                                  controlnet_block_samples, controlnet_single_block_samples, controlnet_blocks_repeat)
        return Transformer2DModelOutput(sample=out) if return_dict else (out,)


NunchakuFluxTransformer2dModel = NunchakuFluxTransformer2DModelV2  # legacy class name (transformer_flux.py)
