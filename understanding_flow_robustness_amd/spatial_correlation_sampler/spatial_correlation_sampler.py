"""Autograd wrapper with the reference's public names and defaults
(Correlation_Module/spatial_correlation_sampler/spatial_correlation_sampler.py:8-147),
bound to the gfx950 backend instead of the pybind module.
"""
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from .. import spatial_correlation_sampler_backend as correlation


class SpatialCorrelationSamplerFunction(Function):
    @staticmethod
    def forward(ctx, input1, input2, kernel_size=1, patch_size=1, stride=1, padding=0, dilation=1,
                dilation_patch=1):
        ctx.save_for_backward(input1, input2)
        ctx.geometry = (*_pair(kernel_size), *_pair(patch_size), *_pair(padding), *_pair(dilation),
                        *_pair(dilation_patch), *_pair(stride))
        return correlation.forward(input1, input2, *ctx.geometry)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input1, input2 = ctx.saved_tensors
        grad_input1, grad_input2 = correlation.backward(input1, input2, grad_output, *ctx.geometry)
        return grad_input1, grad_input2, None, None, None, None, None, None


def spatial_correlation_sample(input1, input2, kernel_size=1, patch_size=1, stride=1, padding=0,
                               dilation=1, dilation_patch=1):
    """out[b,ph,pw,h,w] = sum_c sum_k input1 * shifted input2; every size is an int or a pair."""
    return SpatialCorrelationSamplerFunction.apply(input1, input2, kernel_size, patch_size, stride,
                                                   padding, dilation, dilation_patch)


class SpatialCorrelationSampler(nn.Module):
    def __init__(self, kernel_size=1, patch_size=1, stride=1, padding=0, dilation=1,
                 dilation_patch=1):
        super().__init__()
        self.kernel_size, self.patch_size, self.stride = kernel_size, patch_size, stride
        self.padding, self.dilation, self.dilation_patch = padding, dilation, dilation_patch

    def forward(self, input1, input2):
        return SpatialCorrelationSamplerFunction.apply(
            input1, input2, self.kernel_size, self.patch_size, self.stride, self.padding,
            self.dilation, self.dilation_patch)
