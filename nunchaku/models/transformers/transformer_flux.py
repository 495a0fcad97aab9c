"""reference: nunchaku/models/transformers/transformer_flux.py (legacy class name; same model on MI355X)."""
from nunchaku_amd.models.transformer_flux import NunchakuFluxTransformer2dModel  # noqa: F401
