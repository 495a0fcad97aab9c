"""reference: nunchaku/models/transformers/transformer_qwenimage.py:159-453."""
from nunchaku_amd.models.qwenimage import (  # noqa: F401
    NunchakuQwenImageTransformer2DModel,
    NunchakuQwenImageTransformerBlock,
)
