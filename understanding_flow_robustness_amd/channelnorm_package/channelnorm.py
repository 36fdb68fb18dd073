"""`from channelnorm_package.channelnorm import ChannelNorm` of FlowNet2's model code resolves here."""
from ..warp_ops import ChannelNorm, ChannelNormFunction  # noqa: F401
