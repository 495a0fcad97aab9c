"""nunchaku_amd -- MI355X (gfx950) native SVDQuant W4A4 + low-rank hot path.

Mirrors the operator surface of the reference package ``nunchaku`` for that path:

    nunchaku_amd._C.ops.{quantize_w4a4_act_fuse_lora, gemm_w4a4}   <- nunchaku._C.ops (csrc/pybind.cpp:108-116)
    nunchaku_amd.ops.{quantize, gemm, fused}                        <- nunchaku/ops/*.py
    nunchaku_amd.models.linear.SVDQW4A4Linear                       <- nunchaku/models/linear.py

The compute lives in hand-written HIP kernels behind a C ABI (include/svdq_amd.h); there is no
CPU fallback: importing works anywhere, calling an op without the built library or without a
GPU raises.
"""

__version__ = "0.1.0"
