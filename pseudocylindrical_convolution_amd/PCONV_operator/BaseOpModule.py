"""Common base of the operator modules (reference: PCONV_operator/BaseOpModule.py).

An operator module holds one native op object per GPU id in `self.op` and picks
the one that matches the input's device.  Moving the module (`model.to('cuda:1')`)
re-targets a single-device op, as the reference does in `custom_op_to`
(BaseOpModule.py:33-39); here the target device is read by probing the conversion
function nn.Module hands to `_apply`, not by re-entering `.to()`.
"""
import torch
from torch import nn


class BaseOpModule(nn.Module):

    def __init__(self, devices=0):
        super(BaseOpModule, self).__init__()
        self.device_list = [devices] if isinstance(devices, int) else list(devices)
        self.op = {}

    def _apply(self, fn, *args, **kwargs):
        super(BaseOpModule, self)._apply(fn, *args, **kwargs)
        try:
            target = fn(torch.empty(0)).device
        except Exception:
            target = None
        if target is not None and target.type != 'cpu' and target.index is not None:
            self.custom_op_to(target)
        return self

    def custom_op_to(self, device):
        """re-key a single-device op dict to `device` (BaseOpModule.py:33-39)"""
        if device is None or len(self.op) != 1:
            return
        new_id = device.index
        old_id = next(iter(self.op))
        if new_id is not None and new_id != old_id:
            self.op[new_id] = self.op.pop(old_id)
            self.op[new_id].to(new_id)
            self.device_list = [new_id]

    def custom_op_replicate(self, other):
        other.op = self.op
        return other

    def _replicate_for_data_parallel(self):
        # replicas share the per-GPU op objects (BaseOpModule.py:22-31)
        replica = super(BaseOpModule, self)._replicate_for_data_parallel()
        replica.op = self.op
        return replica

    def native(self, x):
        """the op object that serves tensor x"""
        gid = x.device.index if x.device.index is not None else self.device_list[0]
        try:
            return self.op[gid]
        except KeyError:
            raise RuntimeError('%s has no op for device %s (built for %s)' %
                               (type(self).__name__, x.device, sorted(self.op)))


class _NativeCall(torch.autograd.Function):
    """Bridge between autograd and a stateful native op: forward calls
    `op.<method>(*tensors)`, backward calls `op.backward` where the native side
    has one (the codec itself runs under no_grad)."""

    @staticmethod
    def forward(ctx, module, method, n_out, grad_mode, *tensors):
        prepared = [t.contiguous() if (torch.is_tensor(t) and not t.is_contiguous()) else t for t in tensors]
        first = next(t for t in prepared if torch.is_tensor(t))
        op = module.native(first)
        outs = getattr(op, method)(*prepared)
        ctx.op, ctx.grad_mode, ctx.n_in = op, grad_mode, len(tensors)
        if n_out == 1:
            return outs[0]
        if grad_mode == 'none':
            ctx.mark_non_differentiable(*outs[:n_out])
        return tuple(outs[:n_out])

    @staticmethod
    def backward(ctx, *grads):
        if ctx.grad_mode == 'none':
            return (None,) * (4 + ctx.n_in)
        g = grads[0]
        if not g.is_contiguous():
            g = g.contiguous()
        res = ctx.op.backward(g)
        # one gradient per tensor argument where the op has one (EntropyGmm: weight, delta, mean, label)
        return (None, None, None, None) + tuple(res[i] if i < len(res) else None for i in range(ctx.n_in))


def native_call(module, method, tensors, n_out=1, grad_mode='first'):
    """Run `module.native(x).<method>(*tensors)` through autograd."""
    return _NativeCall.apply(module, method, n_out, grad_mode, *tensors)
