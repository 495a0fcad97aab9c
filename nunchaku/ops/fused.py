"""reference: nunchaku/ops/fused.py:14-79, 82-178."""
from nunchaku_amd.ops.fused import fused_gelu_mlp, fused_qkv_norm_rottary  # noqa: F401
