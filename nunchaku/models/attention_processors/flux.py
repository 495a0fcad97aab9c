"""reference: nunchaku/models/attention_processors/flux.py:14-237."""
from nunchaku_amd.models.attention_processors import NunchakuFluxFA2Processor, NunchakuFluxFP16AttnProcessor  # noqa: F401
