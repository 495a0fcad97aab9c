"""reference: nunchaku/models/transformers/__init__.py."""
from .transformer_flux_v2 import NunchakuFluxTransformer2dModel, NunchakuFluxTransformer2DModelV2  # noqa: F401
from .transformer_qwenimage import NunchakuQwenImageTransformer2DModel  # noqa: F401
