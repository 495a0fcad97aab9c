"""reference: nunchaku/models/linear.py:13-414."""
from nunchaku_amd.models.linear import AWQW4A16Linear, SVDQW4A4Linear  # noqa: F401
