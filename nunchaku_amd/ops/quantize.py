"""Activation quantiser wrapper (reference: nunchaku/ops/quantize.py:11-81)."""

from __future__ import annotations

import torch

from .._C import ops
from ..mode import alloc_lora_act
from ..utils import ceil_divide


def svdq_quantize_w4a4_act_fuse_lora_cuda(
    input: torch.Tensor,
    output: torch.Tensor | None = None,
    oscales: torch.Tensor | None = None,
    lora_down: torch.Tensor | None = None,
    lora_act_out: torch.Tensor | None = None,
    smooth: torch.Tensor | None = None,
    fuse_glu: bool = False,
    fp4: bool = False,
    pad_size: int = 256,
    ln: tuple | None = None,
    pool=None,
    lora_act_zeroed: bool = False,
    second: dict | None = None,
) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """4-bit quantisation of ``input`` [M, K] (``fuse_glu``: [M, 2K] interleaved (value, gate) pairs, quantised as
    ``value * silu(gate)``) plus the low-rank down projection.

    Returns ``(output [M_pad, 3K/4] uint8, oscales [K/64, M_pad], lora_act_out [M_pad, R] float32)``
    with ``M_pad = ceil(M / pad_size) * pad_size``.  ``output`` and ``oscales`` are opaque (the FP6
    operand image / scale image of this library: 6 bits per 4-bit code, so the code buffer is 1.5x the
    reference's ``[M_pad, K/2]``); ``lora_act_out`` holds the true fp32 projection.
    ``ln = (stats, scale, shift)`` (extension): quantise ``layer_norm(input) * scale + shift`` computed on
    the fly from the row statistics of ``ops.elementwise.residual_gate_stats`` -- the AdaLayerNormZero front end.
    """
    if fp4:
        raise NotImplementedError("NVFP4 is not available on MI355X")
    M, K = input.shape
    if fuse_glu:  # [M, 2K] (value, gate) pairs -> K quantised channels (the reference's wrapper sizes its buffers from the
        K //= 2   # input width and can only be called with caller-sized ones in this mode, launch_impl.cuh:463-467)
    R = lora_down.shape[1]
    M_pad = ceil_divide(M, pad_size) * pad_size
    dev = input.device
    if output is None:
        output = torch.empty(M_pad, K * 3 // 4, dtype=torch.uint8, device=dev)
    if oscales is None:
        if K % 128:
            raise ValueError("K must be a multiple of 128")
        oscales = torch.empty(K // 64, M_pad, dtype=input.dtype, device=dev)
    zeroed = bool(lora_act_zeroed)  # only meaningful with a caller-provided lora_act_out
    if lora_act_out is None:
        # a 4th element of ``ln`` is a ZeroPool of fp32 scratch cleared by the preceding residual_gate_stats pass
        if pool is None:
            pool = ln[3] if ln is not None and len(ln) > 3 else None
        lora_act_out, zeroed = alloc_lora_act(M_pad, R, dev, pool)  # fp32, or int64 fixed point in deterministic mode
    if ln is None:
        ops.quantize_w4a4_act_fuse_lora(input, output, oscales, lora_down, lora_act_out, smooth, fuse_glu, fp4,
                                        lora_act_zeroed=zeroed, second=second)
    else:
        ops.quantize_w4a4_act_fuse_lora(input, output, oscales, lora_down, lora_act_out, smooth, fuse_glu, fp4,
                                        ln_stats=ln[0], mod_scale=ln[1], mod_shift=ln[2], lora_act_zeroed=zeroed, second=second)
    return output, oscales, lora_act_out
