"""reference: nunchaku/models/transformers/transformer_flux_v2.py:345-561 (and the legacy class name of transformer_flux.py)."""
from nunchaku_amd.models.transformer_flux import NunchakuFluxTransformer2dModel, NunchakuFluxTransformer2DModelV2  # noqa: F401
