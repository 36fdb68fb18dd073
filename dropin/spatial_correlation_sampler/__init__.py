"""Top-level package the reference imports (`from spatial_correlation_sampler import spatial_correlation_sample`,
models/submodules.py:6, models/PWCNet.py:11, models/raft/corr.py:13): re-export of the gfx950 mirror."""
from understanding_flow_robustness_amd.spatial_correlation_sampler import (  # noqa: F401
    SpatialCorrelationSampler, SpatialCorrelationSamplerFunction, spatial_correlation_sample)
