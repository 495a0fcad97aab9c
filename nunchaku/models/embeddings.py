"""reference: nunchaku/models/embeddings.py (rotary table packing used by the fused QKV epilogue)."""
from nunchaku_amd.models.embeddings import flux_pos_embed, pack_rotemb  # noqa: F401
