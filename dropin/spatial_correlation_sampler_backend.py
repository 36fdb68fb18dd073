"""Top-level name the reference imports (see dropin/README.md): re-export of understanding_flow_robustness_amd.spatial_correlation_sampler_backend."""
from understanding_flow_robustness_amd.spatial_correlation_sampler_backend import *  # noqa: F401,F403
from understanding_flow_robustness_amd.spatial_correlation_sampler_backend import backward, forward  # noqa: F401
