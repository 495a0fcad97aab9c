"""AWQ W4A16 GEMV wrappers (reference: nunchaku/ops/gemv.py:10-57; keyword names are the reference's)."""

from __future__ import annotations

import torch

from .._C import ops


def awq_gemv_w4a16_cuda(in_feats: torch.Tensor, kernel: torch.Tensor, scaling_factors: torch.Tensor, zeros: torch.Tensor,
                        m: int, n: int, k: int, group_size: int = 64, bias: torch.Tensor | None = None,
                        out_chunks: int = 1) -> torch.Tensor:
    """``in_feats`` [m, k] 16-bit, ``kernel`` [n/4, k/2] int32 (checkpoint order), ``scaling_factors`` / ``zeros``
    [k/group_size, n] -> a new [m, n] tensor.  Extensions: ``bias`` fuses the module's 16-bit ``output.add_(bias)``,
    ``out_chunks`` = c writes the output de-interleaved into c contiguous [n/c] vectors."""
    return ops.gemv_awq(in_feats, kernel, scaling_factors, zeros, m, n, k, group_size, bias, out_chunks)


def awq_gemv_w4a16_batched(in_feats: torch.Tensor, layers) -> list[torch.Tensor]:
    """All of ``layers`` (AWQW4A16Linear) applied to the same single row ``in_feats`` in one launch (extension): the
    modulation projections of every block of a denoising step depend only on the timestep embedding."""
    return ops.gemv_awq_batched(in_feats, list(layers))
