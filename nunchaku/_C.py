"""Drop-in for the reference's pybind11 module ``nunchaku._C`` (nunchaku/csrc/pybind.cpp:108-123): submodules ``ops``
(``gemm_w4a4``, ``quantize_w4a4_act_fuse_lora``, ``attention_fp16``, ``gemv_awq``) and ``utils`` (no-op toggles)."""
from nunchaku_amd._C import ops, utils  # noqa: F401
