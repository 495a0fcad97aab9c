"""reference: nunchaku/models/utils.py:52-262 (layer-wise CPU offloading)."""
from nunchaku_amd.models.offload import CPUOffloadManager  # noqa: F401
