"""reference: nunchaku/models/__init__.py."""
from .transformers import (  # noqa: F401
    NunchakuFluxTransformer2dModel,
    NunchakuFluxTransformer2DModelV2,
    NunchakuQwenImageTransformer2DModel,
)
