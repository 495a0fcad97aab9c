"""reference: nunchaku/utils.py (the helpers the hot path uses)."""
from nunchaku_amd.utils import ceil_divide, get_precision, pad_tensor  # noqa: F401
