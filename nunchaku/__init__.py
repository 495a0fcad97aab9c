"""``nunchaku`` import surface on MI355X: the reference's package layout re-exported from ``nunchaku_amd``.

``from nunchaku import NunchakuFluxTransformer2DModelV2`` / ``from nunchaku.models.linear import SVDQW4A4Linear`` /
``from nunchaku._C import ops`` / ``from nunchaku.ops.fused import fused_gelu_mlp`` resolve to the MI355X implementation
(reference: nunchaku/__init__.py:1-17, nunchaku/csrc/pybind.cpp:108-123).  ``nunchaku._C.ops`` takes the reference's
positional signatures, reference-sized opaque buffers and checkpoint-layout parameters (nunchaku_amd/_C.py), so the
reference's own ``ops/*.py`` / ``models/linear.py`` callers run against it unchanged (tests/test_nunchaku_shim.py).
Model families outside the FLUX / Qwen-Image hot path (SANA, Z-Image, T5) are not part of this package.
"""
from .models import (  # noqa: F401
    NunchakuFluxTransformer2dModel,
    NunchakuFluxTransformer2DModelV2,
    NunchakuQwenImageTransformer2DModel,
)

__all__ = ["NunchakuFluxTransformer2dModel", "NunchakuFluxTransformer2DModelV2", "NunchakuQwenImageTransformer2DModel"]
