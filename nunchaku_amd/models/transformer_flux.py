"""diffusers-facing surface of the FLUX transformer.

Reference: ``NunchakuFluxTransformer2DModelV2`` (nunchaku/models/transformers/transformer_flux_v2.py:346-561), a
``diffusers.FluxTransformer2DModel`` subclass that a ``FluxPipeline`` takes as its ``transformer``:

    transformer = NunchakuFluxTransformer2DModelV2.from_pretrained("svdq-int4_r32-flux.1-dev.safetensors")
    pipe = FluxPipeline.from_pretrained("black-forest-labs/FLUX.1-dev", transformer=transformer, torch_dtype=torch.bfloat16)

This adapter keeps that call contract without importing diffusers (it is not a dependency of this package; the
pipeline only duck-types its transformer): keyword ``forward`` with the pipeline's argument names, a ``config`` with the
fields the pipeline reads, ``dtype`` / ``device``, ``return_dict`` handling and an output object with ``.sample``.
The legacy class name ``NunchakuFluxTransformer2dModel`` (transformer_flux.py) is an alias; its LoRA entry points
``update_lora_params`` / ``set_lora_strength`` exist on the model (per-layer factors, see flux.py).
"""

from __future__ import annotations

import json
import os
from types import SimpleNamespace

import torch

from . import loader
from .flux import FluxTransformerAMD


class Transformer2DModelOutput:
    """Stand-in for ``diffusers.models.modeling_outputs.Transformer2DModelOutput`` (one field)."""

    def __init__(self, sample: torch.Tensor):
        self.sample = sample

    def __getitem__(self, i):
        return (self.sample,)[i]


class NunchakuFluxTransformer2DModelV2(FluxTransformerAMD):
    def __init__(self, config: dict | None = None, rank: int = 32, torch_dtype: torch.dtype = torch.bfloat16, device="cuda"):
        cfg = dict(num_layers=19, num_single_layers=38, num_attention_heads=24, attention_head_dim=128, in_channels=64,
                   joint_attention_dim=4096, pooled_projection_dim=768, guidance_embeds=True, axes_dims_rope=(16, 56, 56),
                   patch_size=1, out_channels=None)
        cfg.update(config or {})
        super().__init__(num_layers=cfg["num_layers"], num_single_layers=cfg["num_single_layers"],
                         dim=cfg["num_attention_heads"] * cfg["attention_head_dim"], heads=cfg["num_attention_heads"],
                         in_channels=cfg["in_channels"], joint_attention_dim=cfg["joint_attention_dim"],
                         pooled_projection_dim=cfg["pooled_projection_dim"], rank=rank, guidance_embeds=cfg["guidance_embeds"],
                         axes_dims_rope=tuple(cfg["axes_dims_rope"]), torch_dtype=torch_dtype, device=device)
        self.config = SimpleNamespace(**cfg)  # FluxPipeline reads .in_channels and .guidance_embeds

    @property
    def dtype(self) -> torch.dtype:
        return self.dtype_

    @property
    def device(self) -> torch.device:
        return self.proj_out.weight.device

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str | os.PathLike, **kwargs):
        """A local nunchaku ``.safetensors`` file (reference :373-428; there is no hub access in this package)."""
        from safetensors import safe_open

        if kwargs.get("offload", False):
            raise NotImplementedError("Offload is not supported for FluxTransformer2DModelV2")  # as the reference, :398-399
        path = os.fspath(pretrained_model_name_or_path)
        if not path.endswith((".safetensors", ".sft")):
            raise ValueError("Only safetensors are supported")
        sd = {}
        with safe_open(path, framework="pt", device="cpu") as f:
            meta = f.metadata() or {}
            for k in f.keys():
                sd[k] = f.get_tensor(k)
        qcfg = json.loads(meta.get("quantization_config", "{}"))
        if any(k.endswith(".wcscales") for k in sd):
            raise NotImplementedError("NVFP4 checkpoints need Blackwell's block-scaled mma; use the int4 checkpoint on MI355X")
        model = cls(json.loads(meta.get("config", "{}")), rank=qcfg.get("rank", 32),
                    torch_dtype=kwargs.get("torch_dtype", torch.bfloat16), device=kwargs.get("device", "cuda"))
        return loader.load_flux_state_dict(model, sd)

    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor = None,
                pooled_projections: torch.Tensor = None, timestep: torch.Tensor = None, img_ids: torch.Tensor = None,
                txt_ids: torch.Tensor = None, guidance: torch.Tensor = None, joint_attention_kwargs=None,
                controlnet_block_samples=None, controlnet_single_block_samples=None, return_dict: bool = True,
                controlnet_blocks_repeat: bool = False):
        """The ``FluxPipeline`` call (reference :430-561): ``timestep`` arrives divided by 1000, ids as [T, 3] (a leading
        batch axis, deprecated in diffusers, is dropped as the reference does :505-517)."""
        if controlnet_block_samples is not None or controlnet_single_block_samples is not None:
            raise NotImplementedError("ControlNet residuals are out of scope (SURVEY.md section 8)")
        if txt_ids is not None and txt_ids.ndim == 3:
            txt_ids = txt_ids[0]
        if img_ids is not None and img_ids.ndim == 3:
            img_ids = img_ids[0]
        if self.guidance_embed is not None and guidance is None:
            raise ValueError("this checkpoint has guidance embeddings: pass guidance")
        out = super().forward(hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids, guidance)
        return Transformer2DModelOutput(sample=out) if return_dict else (out,)


NunchakuFluxTransformer2dModel = NunchakuFluxTransformer2DModelV2  # legacy class name (transformer_flux.py)
