"""Mirror of the reference package `spatial_correlation_sampler`
(Correlation_Module/spatial_correlation_sampler/__init__.py:1-4)."""
from .spatial_correlation_sampler import (  # noqa: F401
    SpatialCorrelationSampler,
    SpatialCorrelationSamplerFunction,
    spatial_correlation_sample,
)
