"""Mirror of models/resample2d_package/resample2d.py:7-56 on the gfx950 backend."""
from torch.autograd import Function
from torch.nn.modules.module import Module

from .. import resample2d_cuda


class Resample2dFunction(Function):
    @staticmethod
    def forward(ctx, input1, input2, kernel_size=1, bilinear=True):
        assert input1.is_contiguous()
        assert input2.is_contiguous()
        ctx.save_for_backward(input1, input2)
        ctx.kernel_size, ctx.bilinear = kernel_size, bilinear
        _, d, _, _ = input1.size()
        b, _, h, w = input2.size()
        output = input1.new_empty((b, d, h, w))  # fully written by the kernel
        resample2d_cuda.forward(input1, input2, output, kernel_size, bilinear)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        grad_output = grad_output.contiguous()
        input1, input2 = ctx.saved_tensors
        grad_input1 = input1.new_empty(input1.size())  # zeroed + scattered by the C ABI call
        grad_input2 = input1.new_empty(input2.size())
        resample2d_cuda.backward(input1, input2, grad_output, grad_input1, grad_input2,
                                 ctx.kernel_size, ctx.bilinear)
        return grad_input1, grad_input2, None, None


class Resample2d(Module):
    def __init__(self, kernel_size=1, bilinear=True):
        super().__init__()
        self.kernel_size, self.bilinear = kernel_size, bilinear

    def forward(self, input1, input2):
        return Resample2dFunction.apply(input1.contiguous(), input2, self.kernel_size, self.bilinear)
