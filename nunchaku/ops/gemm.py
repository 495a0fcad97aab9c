"""reference: nunchaku/ops/gemm.py:12-160."""
from nunchaku_amd.ops.gemm import svdq_gemm_w4a4_cuda  # noqa: F401
