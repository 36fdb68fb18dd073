"""`from resample2d_package.resample2d import Resample2d` of FlowNet2's model code resolves here."""
from ..warp_ops import Resample2d, Resample2dFunction  # noqa: F401
