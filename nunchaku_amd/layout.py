"""Load-time re-layout: reference checkpoint tensors (NVIDIA mma fragment order,
nunchaku/lora/flux/packer.py:187-301,362-437) -> the CDNA4 orders the HIP kernels consume.

Each function returns a NEW tensor of the same shape/dtype holding the permuted data; the GPU does
the work (csrc/repack.hip through the C ABI).  ``SVDQW4A4Linear.repack_()`` applies them in place.
"""

from __future__ import annotations

import torch

from . import _lib


def _run(fn_name, src: torch.Tensor, *dims) -> torch.Tensor:
    lib = _lib.load()
    if not src.is_cuda:
        raise RuntimeError("nunchaku_amd.layout: tensors must be on the GPU (no CPU path)")
    src = src.contiguous()
    dst = torch.empty_like(src)
    rc = getattr(lib, fn_name)(src.data_ptr(), dst.data_ptr(), *dims, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, fn_name)
    return dst


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return _lib.SVDQ_BF16
    if t.dtype == torch.float16:
        return _lib.SVDQ_FP16
    raise ValueError(f"nunchaku_amd.layout: weight scales must be bfloat16 or float16, not {t.dtype}")


def act_image_shape(rows: int, K: int) -> tuple[int, int]:
    """Shape (uint8) of the FP6 operand image of a [rows, K] matrix of 4-bit codes (6 bits per code)."""
    if K % 128 or rows % 32:
        raise ValueError("FP6 operand images need K % 128 == 0 and rows % 32 == 0")
    return rows, K * 3 // 4


def repack_qweight(qweight: torch.Tensor) -> torch.Tensor:
    """[N, K/2] int8 reference order -> [N, 3K/4] FP6 operand image (csrc/svdq_common.h "F6")."""
    lib = _lib.load()
    if not qweight.is_cuda:
        raise RuntimeError("nunchaku_amd.layout: tensors must be on the GPU (no CPU path)")
    qweight = qweight.contiguous()
    N, Kh = qweight.shape
    dst = torch.empty(act_image_shape(N, Kh * 2), dtype=torch.int8, device=qweight.device)
    rc = lib.svdq_repack_qweight(qweight.data_ptr(), dst.data_ptr(), N, Kh * 2, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "svdq_repack_qweight")
    return dst


def repack_wscales(wscales: torch.Tensor) -> torch.Tensor:
    """[K/64, N] 16-bit reference order -> scale image [N/32][K/128][2][32] (same storage shape)."""
    G, N = wscales.shape
    # ABI 21: the image holds 32 x the scale (the GEMM's product MFMA runs without MX block scales: csrc/repack.hip).  Exact in bf16; fp16 tops out at 2047
    if wscales.dtype == torch.float16 and wscales.numel() and float(wscales.abs().max()) > 2047.0:
        raise ValueError("repack_wscales: an fp16 weight scale above 2047 cannot be held in the kernel's scale image (32 x scale); use bfloat16")
    return _run("svdq_repack_wscales", wscales, G, N, _dt(wscales))


def repack_vec(v: torch.Tensor) -> torch.Tensor:
    """bias / smooth_factor [N] reference order -> natural."""
    return _run("svdq_repack_vec", v, v.numel())


def repack_lowrank(w: torch.Tensor, down: bool) -> torch.Tensor:
    """proj_up [N, R] -> natural [n][r];  proj_down [K, R] -> rank-major [r][k] (same storage shape)."""
    C_, R = w.shape
    return _run("svdq_repack_lowrank", w, C_, R, 1 if down else 0)


# ---- inverses: kernel layout -> the reference's checkpoint layout (state_dict(), host offload) -----------------------------
def unrepack_qweight(img: torch.Tensor) -> torch.Tensor:
    """[N, 3K/4] FP6 operand image -> [N, K/2] int8 in the reference order (exact inverse of :func:`repack_qweight`)."""
    lib = _lib.load()
    if not img.is_cuda:
        raise RuntimeError("nunchaku_amd.layout: tensors must be on the GPU (no CPU path)")
    img = img.contiguous()
    N, B = img.shape
    K = B * 4 // 3
    dst = torch.empty(N, K // 2, dtype=torch.int8, device=img.device)
    _lib.check(lib.svdq_unrepack_qweight(img.data_ptr(), dst.data_ptr(), N, K, torch.cuda.current_stream().cuda_stream), "svdq_unrepack_qweight")
    return dst


def unrepack_wscales(simg: torch.Tensor) -> torch.Tensor:
    G, N = simg.shape
    return _run("svdq_unrepack_wscales", simg, G, N, _dt(simg))


def unrepack_vec(v: torch.Tensor) -> torch.Tensor:
    return _run("svdq_unrepack_vec", v, v.numel())


def unrepack_lowrank(w: torch.Tensor, down: bool) -> torch.Tensor:
    C_, R = w.shape
    return _run("svdq_unrepack_lowrank", w, C_, R, 1 if down else 0)


def unpack_act(act: torch.Tensor, K: int, unsigned: bool = False) -> torch.Tensor:
    """Opaque packed activations -> int8 codes [M_pad, K] (test/debug helper)."""
    lib = _lib.load()
    M_pad = act.numel() * 4 // (3 * K)
    codes = torch.empty(M_pad, K, dtype=torch.int8, device=act.device)
    rc = lib.svdq_unpack_act(act.data_ptr(), codes.data_ptr(), M_pad, K, int(unsigned),
                             torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "svdq_unpack_act")
    return codes


def unpack_scales(simg: torch.Tensor, rows: int) -> torch.Tensor:
    """Opaque scale image -> natural [K/64, rows] (test/debug helper)."""
    lib = _lib.load()
    G = simg.numel() // rows
    nat = torch.empty(G, rows, dtype=simg.dtype, device=simg.device)
    rc = lib.svdq_unpack_scales(simg.data_ptr(), nat.data_ptr(), rows, G, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "svdq_unpack_scales")
    return nat
