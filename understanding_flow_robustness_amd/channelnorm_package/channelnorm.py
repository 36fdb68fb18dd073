"""Mirror of models/channelnorm_package/channelnorm.py:6-38 on the gfx950 backend."""
from torch.autograd import Function
from torch.nn.modules.module import Module

from .. import channelnorm_cuda


class ChannelNormFunction(Function):
    @staticmethod
    def forward(ctx, input1, norm_deg=2):
        assert input1.is_contiguous()
        b, _, h, w = input1.size()
        output = input1.new_empty((b, 1, h, w))
        channelnorm_cuda.forward(input1, output, norm_deg)
        ctx.save_for_backward(input1, output)
        ctx.norm_deg = norm_deg
        return output

    @staticmethod
    def backward(ctx, grad_output):
        input1, output = ctx.saved_tensors
        grad_input1 = input1.new_empty(input1.size())
        channelnorm_cuda.backward(input1, output, grad_output.contiguous(), grad_input1, ctx.norm_deg)
        return grad_input1, None


class ChannelNorm(Module):
    def __init__(self, norm_deg=2):
        super().__init__()
        self.norm_deg = norm_deg

    def forward(self, input1):
        return ChannelNormFunction.apply(input1, self.norm_deg)
