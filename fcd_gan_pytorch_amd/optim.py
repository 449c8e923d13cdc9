"""Adam / RMSprop for the FCD-GAN nets, one fused HIP launch per network
(csrc/optim.hip) over FLAT parameter / gradient buffers.

Same update rules and defaults as ``torch.optim.Adam`` / ``torch.optim.RMSprop``
as the demos construct them (Demo_USSS.py:121-122, Demo_RSSS.py:151-158,
Demo_WSSS.py:116-122) and the same ``param_groups[i]['lr']`` knob that
``adjust_learning_rate`` (CommonFunc.py:23-37) writes.

Flat layout: at construction every parameter is re-pointed at a slice of one
contiguous fp32 buffer, and its ``.grad`` at the matching slice of one gradient
buffer, so (a) the optimizer step is a single kernel, (b) the data-parallel
gradient exchange is a single RCCL all-reduce per network (``allreduce_grads``),
with the 1/world_size scaling folded into the update kernel.
"""
import torch
import torch.distributed as dist

from . import _ops as ops


class _FlatOptimizer:
    def __init__(self, params, lr):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError('optimizer got an empty parameter list')
        dev = self.params[0].device
        # (the flat-buffer / all-reduce host logic also works on CPU tensors -- used by the gloo
        # tests -- but step() is a HIP kernel and refuses them)
        n = sum(p.numel() for p in self.params)
        self.flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.flat_p[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[off:off + k].view(p.shape)
                p.grad = self.flat_g[off:off + k].view(p.shape)
                off += k
        ops.invalidate_packs(self.params)
        self.param_groups = [{'params': self.params, 'lr': lr}]
        self.steps = 0
        self.grad_scale = 1.0
        self.pre_step_hooks = []

    def zero_grad(self, set_to_none=False):
        # in place: gradients must stay views of the flat buffer (re-point first, WITHOUT copying: whatever a
        # detached .grad holds is exactly what zero_grad is meant to discard)
        self._bind_grads(copy=False)
        self.flat_g.zero_()

    def _bind_grads(self, copy=True):
        """Make every ``p.grad`` the matching view of ``flat_g`` again.  A gradient that was detached from the
        flat buffer (``net.zero_grad()`` with its set_to_none default, ``p.grad = None``, a backward that ran
        while ``.grad`` was None) is copied in first when ``copy`` (a None gradient contributes zeros), so that
        the fused update never steps on stale numbers."""
        lo = self.flat_g.data_ptr()
        off = 0
        for p in self.params:
            k = p.numel()
            g = p.grad
            if g is None or g.data_ptr() != lo + 4 * off or not g.is_contiguous():
                view = self.flat_g[off:off + k].view(p.shape)
                if copy:
                    if g is None:
                        view.zero_()
                    else:
                        view.copy_(g)
                p.grad = view
            off += k

    def allreduce_grads(self, group=None):
        """Data-parallel exchange: one all-reduce(sum) of the flat gradient buffer over
        RCCL (xGMI); the mean is applied inside the next ``step`` (grad_scale)."""
        self._bind_grads()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=group)
            self.grad_scale = 1.0 / dist.get_world_size(group)
        else:
            self.grad_scale = 1.0

    def _require_device(self):
        if self.flat_p.device.type != 'cuda':
            raise ops._lib.FcdError('fcd optimizers step on a CUDA/ROCm device only (call net.to(device) before '
                                    'constructing the optimizer); there is no CPU fallback')

    def _before_step(self):
        self._require_device()
        self._bind_grads()
        for hook in self.pre_step_hooks:      # e.g. the parity tests snapshot flat_g here
            hook(self)

    def _after_step(self):
        self.steps += 1
        self.grad_scale = 1.0
        # the kernel wrote the parameters through raw pointers: tell autograd (a backward over a graph that
        # saved the old weights now raises, as it would after torch.optim's in-place update) and drop the
        # packed / transformed filter copies
        torch._C._increment_version(self.params)
        ops.invalidate_packs(self.params)

    @property
    def lr(self):
        return float(self.param_groups[0]['lr'])


class Adam(_FlatOptimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, lr)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)

    @torch.no_grad()
    def step(self):
        self._before_step()
        ops.adam_step(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0],
                      self.betas[1], self.eps, self.weight_decay, self.steps + 1, self.grad_scale)
        self._after_step()


class RMSprop(_FlatOptimizer):
    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0.0):
        super().__init__(params, lr)
        self.alpha, self.eps, self.weight_decay = alpha, eps, weight_decay
        self.square_avg = torch.zeros_like(self.flat_p)

    @torch.no_grad()
    def step(self):
        self._before_step()
        ops.rmsprop_step(self.flat_p, self.flat_g, self.square_avg, self.lr, self.alpha, self.eps,
                         self.weight_decay, self.grad_scale)
        self._after_step()


def adjust_learning_rate(optimizer, epoch, lr_start=1e-4, lr_max=1e-3, lr_min=1e-6, lr_warm_up_epoch=20,
                         lr_sustain_epochs=0, lr_exp_decay=0.8):
    """Linear warm-up -> sustain -> exponential decay (reference CommonFunc.py:23-37)."""
    if epoch < lr_warm_up_epoch:
        lr = (lr_max - lr_start) / lr_warm_up_epoch * epoch + lr_start
    elif epoch < lr_warm_up_epoch + lr_sustain_epochs:
        lr = lr_max
    else:
        lr = (lr_max - lr_min) * lr_exp_decay ** (epoch - lr_warm_up_epoch - lr_sustain_epochs) + lr_min
    for group in optimizer.param_groups:
        group['lr'] = lr
    return lr
