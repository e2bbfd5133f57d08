"""LowerBound: max(x, bound) whose gradient also passes when it pushes x back
above the bound (reference: PCONV_operator/GDN.py:6-22)."""
import torch
from torch.autograd import Function


class LowerBound(Function):

    @staticmethod
    def forward(ctx, inputs, bound):
        floor = torch.ones_like(inputs) * bound.to(inputs.device) if torch.is_tensor(bound) \
            else torch.full_like(inputs, bound)
        ctx.save_for_backward(inputs, floor)
        return torch.max(inputs, floor)

    @staticmethod
    def backward(ctx, grad_output):
        inputs, floor = ctx.saved_tensors
        keep = (inputs >= floor) | (grad_output < 0)
        return keep.type(grad_output.dtype) * grad_output, None
