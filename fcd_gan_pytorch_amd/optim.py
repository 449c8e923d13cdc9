"""Adam / RMSprop for the FCD-GAN nets, one fused HIP launch per network
(csrc/optim.hip) over FLAT parameter / gradient buffers.

Same update rules and defaults as ``torch.optim.Adam`` / ``torch.optim.RMSprop``
as the demos construct them (Demo_USSS.py:121-122, Demo_RSSS.py:151-158,
Demo_WSSS.py:116-122) and the same ``param_groups[i]['lr']`` knob that
``adjust_learning_rate`` (CommonFunc.py:23-37) writes.

Flat layout: at construction every parameter is re-pointed at a slice of one
contiguous fp32 buffer, and its ``.grad`` at the matching slice of one gradient
buffer, so (a) the optimizer step is a single kernel, (b) the data-parallel
gradient exchange works on contiguous slices of that buffer with the
1/world_size scaling folded into the update kernel.

Data-parallel exchange (collective 1): the flat gradient buffer is
cut into buckets in REVERSE registration order (the Segmentor's decoder first:
``up1.conv.double_conv.0.weight`` alone is 75.5 MB), and ``begin_overlap()`` arms
per-parameter post-accumulate hooks for ONE backward pass: as soon as every
parameter of the next bucket in line has its gradient, that slice is all-reduced
(sum) asynchronously -- on RCCL's own stream, behind the kernels that produced
it -- while the backward pass keeps computing the encoder's gradients.
``allreduce_grads()`` flushes what is left and makes the compute stream wait for
the collectives; the update kernel then divides by the world size.  Buckets are
issued strictly in bucket order, so every rank launches the same sequence of
collectives.  Over xGMI (point-to-point links, ring collectives per-link bound)
few large buckets beat many small ones: default 32 MB.
"""
import torch
import torch.distributed as dist

from . import _ops as ops
from . import dp as _dp


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
        self.epoch = 0                 # bumped by zero_grad: a gradient slot is handed out once per epoch (see grad_slot)
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.flat_p[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[off:off + k].view(p.shape)
                p.grad = self.flat_g[off:off + k].view(p.shape)
                p._fcd_slot = (self, off, -1)      # (owner, offset into flat_g, epoch the slot was last handed out in)
                off += k
        ops.invalidate_packs(self.params)
        self.param_groups = [{'params': self.params, 'lr': lr}]
        self.steps = 0
        self.grad_scale = 1.0
        self.pre_step_hooks = []
        self.bucket_bytes = 32 << 20
        self._buckets = None          # [(lo, hi, n_params)] over flat_g, bucket 0 = LAST parameters
        self._armed = False
        self.last_exchange = None     # diagnostics of the most recent exchange (tests, bench)
        self._hyper = None            # device scalars of the update kernel (use_device_hyper: for callers that capture the step into a hipGraph)

    def zero_grad(self, set_to_none=False):
        """One memset of the flat buffer; every ``.grad`` becomes None.  The backward kernels of the fcd ops then write
        each parameter gradient STRAIGHT into its slice of the flat buffer (``grad_slot``) and hand autograd a view of
        it, which AccumulateGrad adopts as ``.grad`` without a copy or an add -- no per-parameter accumulate kernel.
        A parameter that collects a second gradient in the same epoch (used twice in one graph, or a second backward)
        falls back to autograd's ordinary accumulation; ``_bind_grads`` moves whatever did not land in the flat
        buffer there before the exchange / the update."""
        self.epoch += 1
        for p in self.params:
            p.grad = None
        self.flat_g.zero_()

    def gather_grads(self):
        """Make ``flat_g`` hold every parameter's current gradient (call before reading it directly)."""
        self._bind_grads()
        return self.flat_g

    def grad_slot(self, p):
        """Fresh view of ``p``'s slice of the flat gradient buffer if it may be written directly (first gradient of this
        epoch, nothing accumulated yet), else None."""
        owner, off, used = p._fcd_slot
        if p.grad is not None or used == self.epoch:
            return None
        p._fcd_slot = (owner, off, self.epoch)
        return self.flat_g[off:off + p.numel()].view(p.shape)

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

    # ------------------------------------------------------------- data-parallel exchange
    @staticmethod
    def _world(group=None):
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(group)
        return 1

    def _build_buckets(self):
        """Contiguous slices of flat_g, walking the parameters from the LAST registered one backwards (the order in
        which a backward pass produces their gradients), closing a bucket once it holds >= bucket_bytes."""
        offs, off = [], 0
        for p in self.params:
            offs.append(off)
            off += p.numel()
        self._param_off = offs
        buckets, owner = [], {}
        hi, cnt, cur = off, 0, off
        for i in range(len(self.params) - 1, -1, -1):
            cur = offs[i]
            cnt += 1
            owner[i] = len(buckets)
            if 4 * (hi - cur) >= self.bucket_bytes or i == 0:
                buckets.append((cur, hi, cnt))
                hi, cnt = cur, 0
        self._buckets, self._bucket_of = buckets, owner
        self._index = {id(p): i for i, p in enumerate(self.params)}
        for p in self.params:
            if p.requires_grad and not getattr(p, '_fcd_overlap_hook', None):
                p._fcd_overlap_hook = p.register_post_accumulate_grad_hook(self._on_grad_ready)

    def begin_overlap(self, group=None):
        """Arm the bucketed exchange for the NEXT backward pass (exactly one ``backward()`` must produce this
        net's gradients before ``allreduce_grads``).  No-op on a single rank (unless ``dp.force_exchange``)."""
        self._armed = False
        if not _dp.exchanging(group):
            return False
        if self._buckets is None:
            self._build_buckets()
        self._group = group
        self._pending = [b[2] for b in self._buckets]
        self._next = 0
        self._works = []
        self._armed = True
        return True

    def _launch_ready_buckets(self, force=False):
        while self._next < len(self._buckets) and (force or self._pending[self._next] <= 0):
            lo, hi, _ = self._buckets[self._next]
            self._works.append(dist.all_reduce(self.flat_g[lo:hi], op=dist.ReduceOp.SUM, group=self._group, async_op=True))
            self._launched_early += 0 if force else 1
            self._next += 1

    def _on_grad_ready(self, p):
        if not self._armed:
            return
        i = self._index[id(p)]
        g, off = p.grad, self._param_off[i]
        if g is None or g.data_ptr() != self.flat_g.data_ptr() + 4 * off:
            # a gradient that did not land in the flat buffer (``.grad`` was None when backward ran): move it there
            view = self.flat_g[off:off + p.numel()].view(p.shape)
            if g is not None:
                view.copy_(g)
            p.grad = view
        b = self._bucket_of[i]
        if b < self._next:
            # a second backward pass reached a parameter whose bucket already left (begin_overlap arms exactly ONE pass)
            raise RuntimeError('fcd optimizer: gradient of parameter %d arrived after its bucket was all-reduced; call '
                               'begin_overlap() once per backward pass, or allreduce_grads() without it' % i)
        self._pending[b] -= 1
        self._launch_ready_buckets()

    _launched_early = 0

    def allreduce_grads(self, group=None):
        """Data-parallel exchange: all-reduce(sum) of the flat gradient buffer over RCCL (xGMI); the mean is applied
        inside the next ``step`` (grad_scale).  After ``begin_overlap`` the buckets that became ready during the
        backward pass are already in flight: flush the rest, then make the compute stream wait for all of them.
        Without it: one all-reduce of the whole buffer."""
        world = _dp.exchanging(group)
        self._bind_grads()
        if not world:
            self._armed = False
            self.grad_scale = 1.0
            return
        if self._armed:
            early = self._launched_early
            self._launch_ready_buckets(force=True)
            for w in self._works:
                w.wait()                     # NCCL: stream-level wait; gloo: blocks the host
            self.last_exchange = dict(buckets=len(self._buckets), launched_during_backward=early,
                                      bytes=[4 * (hi - lo) for lo, hi, _ in self._buckets])
            self._launched_early = 0
            self._works = []
            self._armed = False
        else:
            dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=group)
            self.last_exchange = dict(buckets=1, launched_during_backward=0, bytes=[4 * self.flat_g.numel()])
        self.grad_scale = 1.0 / world

    def broadcast_state(self, src=0, group=None):
        """Start-of-training weight sync (SURVEY.md 8e "weights broadcast from rank 0"): parameters as ONE flat
        buffer; optimizer moments are zeros on every rank at that point and need no exchange."""
        if _dp.exchanging(group):
            dist.broadcast(self.flat_p, src, group=group)
            torch._C._increment_version(self.params)
            ops.invalidate_packs(self.params)

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
        ops.refresh_packs(self.params)          # F(4x4) packs re-packed in place by one launch, the others dropped

    @property
    def lr(self):
        return float(self.param_groups[0]['lr'])

    # ---- for callers that capture the step into a hipGraph: the update kernel reads its per-step scalars from device memory
    def hyper_values(self, step):
        """Floats the ``_h`` kernel reads for the ``step``-th (1-based) update."""
        return [self.lr]

    def use_device_hyper(self):
        """From now on ``step()`` launches the ``_h`` kernel (scalars from ``self._hyper``); call :meth:`write_hyper` before
        every step / replay."""
        if self._hyper is None:
            self._hyper = torch.zeros(4, dtype=torch.float32, device=self.flat_p.device)
        return self._hyper

    def _refresh_hyper(self):
        """Eager ``step()`` in device-hyper mode writes the current scalars itself; while a hipGraph is being captured the
        host-to-device write is not capturable (and not wanted: the replaying caller writes before every replay)."""
        if not torch.cuda.is_current_stream_capturing():
            self.write_hyper()

    def write_hyper(self, step=None):
        vals = self.hyper_values(self.steps + 1 if step is None else step)
        # (from pageable memory on purpose: the copy is staged before the call returns, so the next step's values cannot
        #  overtake a copy that is still queued behind earlier replays)
        self._hyper[:len(vals)].copy_(torch.tensor(vals, dtype=torch.float32))


class Adam(_FlatOptimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, lr)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)

    @torch.no_grad()
    def step(self):
        self._before_step()
        if self._hyper is not None:
            self._refresh_hyper()
            ops.adam_step_h(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self._hyper, self.betas[0], self.betas[1],
                            self.eps, self.weight_decay, self.grad_scale)
        else:
            ops.adam_step(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0],
                          self.betas[1], self.eps, self.weight_decay, self.steps + 1, self.grad_scale)
        self._after_step()

    def hyper_values(self, step):
        # exactly what fcd_adam_step derives on the host from (beta1, beta2, step): double precision, rounded to float
        import math
        import numpy as np
        b1, b2 = float(np.float32(self.betas[0])), float(np.float32(self.betas[1]))      # the kernel entry receives them as floats
        return [self.lr, 1.0 - math.pow(b1, step), math.sqrt(1.0 - math.pow(b2, step))]


class RMSprop(_FlatOptimizer):
    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0.0):
        super().__init__(params, lr)
        self.alpha, self.eps, self.weight_decay = alpha, eps, weight_decay
        self.square_avg = torch.zeros_like(self.flat_p)

    @torch.no_grad()
    def step(self):
        self._before_step()
        if self._hyper is not None:
            self._refresh_hyper()
            ops.rmsprop_step_h(self.flat_p, self.flat_g, self.square_avg, self._hyper, self.alpha, self.eps, self.weight_decay,
                               self.grad_scale)
        else:
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
