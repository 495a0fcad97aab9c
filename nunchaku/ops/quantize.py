"""reference: nunchaku/ops/quantize.py:11-81."""
from nunchaku_amd.ops.quantize import svdq_quantize_w4a4_act_fuse_lora_cuda  # noqa: F401
