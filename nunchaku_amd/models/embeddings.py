"""Rotary position tables for FLUX and their packing for the fused QKV epilogue
(reference: nunchaku/models/embeddings.py:9-138)."""

from __future__ import annotations

import torch


def rope(pos: torch.Tensor, dim: int, theta: int) -> torch.Tensor:
    """pos [B, T] -> [B, T, dim/2, 1, 2] float32 holding (sin, cos) of pos * theta^(-2i/dim)."""
    if dim % 2:
        raise ValueError("rope dimension must be even")
    expo = torch.arange(0, dim, 2, dtype=torch.float64, device=pos.device) / dim
    freqs = pos.to(torch.float64)[..., None] * (1.0 / (float(theta) ** expo))
    table = torch.stack([freqs.sin(), freqs.cos()], dim=-1)
    return table.reshape(pos.shape[0], -1, dim // 2, 1, 2).float()


def flux_pos_embed(ids: torch.Tensor, axes_dim=(16, 56, 56), theta: int = 10000) -> torch.Tensor:
    """ids [T, n_axes] -> [1, T, sum(axes_dim)/2, 1, 2] (NunchakuFluxPosEmbed.forward, :73-97)."""
    ids = ids[None, ...]
    parts = [rope(ids[..., i], axes_dim[i], theta) for i in range(ids.shape[-1])]
    return torch.cat(parts, dim=-3)


def pack_rotemb(rotemb: torch.Tensor) -> torch.Tensor:
    """[B, M, D/2, 1, 2] float32 -> [B, M, D] in the order the QKV epilogue reads
    (16-row x 8-float tiles laid out as the m16n8 accumulator fragment; reference :100-138).
    The HIP epilogue decodes exactly this order, so reference-side callers need no change."""
    if rotemb.dtype != torch.float32:
        raise ValueError("rotary table must be float32")
    B, M = rotemb.shape[0], rotemb.shape[1]
    D = rotemb.shape[2] * 2
    if M % 16 or D % 8:
        raise ValueError("M must be a multiple of 16 and D of 8")
    t = rotemb.reshape(B, M // 16, 2, 8, D // 8, 4, 2)  # [B, mt, row_half, row8, d8, pair, sincos]
    t = t.permute(0, 1, 4, 3, 5, 2, 6)  # [B, mt, d8, row8, pair, row_half, sincos]
    return t.contiguous().view(B, M, D)
