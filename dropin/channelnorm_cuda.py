"""Top-level name the reference imports (see dropin/README.md): re-export of understanding_flow_robustness_amd.channelnorm_cuda."""
from understanding_flow_robustness_amd.channelnorm_cuda import *  # noqa: F401,F403
from understanding_flow_robustness_amd.channelnorm_cuda import backward, forward  # noqa: F401
