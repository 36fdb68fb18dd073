"""Top-level name the reference imports (see dropin/README.md): re-export of understanding_flow_robustness_amd.alt_cuda_corr."""
from understanding_flow_robustness_amd.alt_cuda_corr import *  # noqa: F401,F403
from understanding_flow_robustness_amd.alt_cuda_corr import backward, forward  # noqa: F401
