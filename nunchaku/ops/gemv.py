"""reference: nunchaku/ops/gemv.py."""
from nunchaku_amd.ops.gemv import awq_gemv_w4a16_cuda  # noqa: F401
