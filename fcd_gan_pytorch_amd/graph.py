"""Train steps replayed from a hipGraph.

One Demo_RSSS iteration is ~950 kernel launches issued from Python through ctypes and the autograd engine: measured on the
MI355X box (bench.py ``host`` block) the host needs 80 - 85 ms to QUEUE a step whose kernels run 89 ms -- the step is one
scheduling hiccup away from being launch-bound, and eight ranks share the node's 16 usable host cores.  The step is static
(fixed tile shapes, no host decision depends on device data: the skip-if-empty rules of Loss.py:115-119 / :133-139 are
evaluated on the device), so it is captured ONCE into a hipGraph -- forward, both backward passes, gradient exchange,
optimizer kernels, BatchNorm running-statistic updates, filter re-packing -- and replayed: the host cost of a step becomes one
graph launch plus a few scalar writes.

    step = graph.GraphedStep(steps.rsss_adversarial_step, nets=(netS, netD, netG, crit), optimizers=(optS, optD))
    out = step(netS, netD, netG, crit, optS, optD, x, y, region)       # same call as the plain function

What a replay cannot carry through recorded launch arguments goes through device memory: the optimizers' learning rate /
Adam bias corrections (``optim._FlatOptimizer.use_device_hyper``: written before every replay, so
``adjust_learning_rate`` keeps working), and the input tiles (copied into the captured buffers).  Python-side bookkeeping the
replay skips is redone after it: optimizer step counters, parameter version counters, packed-filter caches.

Streams: every call -- eager warm-up, capture, replay -- is issued on ONE side stream the wrapper owns, ordered against the
caller's stream on entry and exit.  Autograd ties each parameter's gradient-accumulation node to the stream it was created on;
capturing on another stream than the one the warm-up steps ran on would make the engine synchronise the capturing stream with
a non-capturing one (an invalid capture).  Returned tensors are detached: nothing keeps the previous step's autograd graph (and
its accumulation nodes) alive into the capture.

Data parallelism: NOT captured.  Measured on the MI355X box (ROCm 7.0 / PyTorch 2.10, one-rank ``nccl`` group with
``dp.force_exchange``): with the bucketed all-reduces recorded in the graph the replays are bit-identical, but in 3 of 5 runs the
process group's WATCHDOG THREAD -- which polls the events of earlier, eagerly issued collectives -- hit
``hipErrorStreamCaptureUnsupported`` while the stream was capturing and aborted the process.  A wrapper whose process group is
exchanging therefore never captures: it calls the step launch by launch (``self.refused_dp``).  Nothing is lost: at one rank the
replayed step is not faster either (same box, 90.8 / 91.3 ms replayed vs 90.4 / 90.6 ms launch by launch at the headline
workload; the step is kernel-bound, its ~950 launches cost the host 10 - 12 ms, and ``hipGraphLaunch`` spends as long enqueuing
the nodes) -- the wrapper is for hosts that ARE launch-bound (fewer cores per rank, smaller tiles).

Rules: tensors in the returned dict are the graph's own buffers -- valid until the next call; tensor arguments may be passed
positionally or by keyword (both are copied into the graph's input buffers before a replay); a step function that calls one
optimizer's ``step()`` twice is not captured (``self.refused_multistep``); a call whose tensor shapes or
keyword arguments differ from every captured signature is captured separately (``max_graphs``) or runs eagerly; the first
``warmup`` calls run eagerly (kernel modules, allocator pools and packed frozen filters settle).  Anything the graph reads
that lives outside it (frozen VGG / Generator filters and their packed forms) is kept alive by the wrapper and its version is
checked before every replay: editing a frozen weight re-captures.
"""
import torch

from . import _ops as ops
from . import dp as _dp


def _tsig(a):
    return (tuple(a.shape), str(a.dtype), str(a.device))


def _sig(args, kw):
    """Signature a captured graph is valid for: shapes / dtypes / devices of every tensor -- positional AND keyword (a tensor
    passed as ``region=...`` is an input like any other: it gets a static buffer and is refreshed before every replay) -- plus the
    identity of the other positional arguments and the values of the other keyword arguments."""
    out = [(_tsig(a) if torch.is_tensor(a) else ('obj', id(a))) for a in args]
    kws = tuple(sorted((k, (('tensor',) + _tsig(v)) if torch.is_tensor(v) else ('value', v)) for k, v in kw.items()))
    return tuple(out), kws


class GraphedStep:
    def __init__(self, fn, nets=(), optimizers=(), warmup=2, max_graphs=2):
        self.fn, self.nets, self.optimizers = fn, tuple(nets), tuple(optimizers)
        self.warmup, self.max_graphs = int(warmup), int(max_graphs)
        self.calls = 0
        self._graphs = {}
        self.replays = 0
        self.eager_calls = 0
        self.enabled = True
        self.refused_dp = False
        self.refused_multistep = False     # the step function calls one optimizer's step() more than once: not captured (see _capture)
        self._side = None

    # -------------------------------------------------------------------------------------------------
    def _frozen_state(self):
        """(tensor, version) of every parameter / buffer of the nets that no captured optimizer owns, plus everything the fcd
        caches hang on modules and tensors (packed filters, folded conv+BN filters): kept alive for the graph's lifetime."""
        owned = {id(p) for o in self.optimizers for p in o.params}
        watch, keep = [], []
        for net in self.nets:
            for t in list(net.parameters()) + list(net.buffers()):
                if id(t) not in owned and not getattr(t, '_fcd_graph_mutable', False):
                    watch.append(t)
                pk = t.__dict__.get('_fcd_pack') if hasattr(t, '__dict__') else None
                if pk:
                    keep.append(dict(pk))
            for m in net.modules():
                for k, v in m.__dict__.items():
                    if k.startswith('_fcd_'):
                        keep.append(v)
        return watch, keep

    def _capture(self, args, kw):
        dev = next(a for a in args if torch.is_tensor(a)).device
        static = [a.clone() if torch.is_tensor(a) else a for a in args]
        static_kw = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in kw.items()}
        counts = {}
        for o in self.optimizers:                      # count the step() calls of each optimizer inside one fn call
            o.use_device_hyper()
            o.write_hyper()
            # every packed / transformed form of a TRAINED filter must be produced by a kernel recorded in the graph: a pack that
            # happens to be current now (the net ran after its last update, e.g. the Discriminator inside the Segmentor step) would
            # be read as a constant by every replay
            ops.invalidate_packs(o.params)

            def counted(o=o):
                # capture records the update launch, nothing runs: no step / version bookkeeping here (the replay does it); only
                # the packed-filter caches must go, so that a later forward inside the same step re-packs the UPDATED weights
                counts[id(o)] = counts.get(id(o), 0) + 1
                o.grad_scale = 1.0
                ops.invalidate_packs(o.params)
            o._after_step = counted
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=self._side):
                out = self._detached(self.fn(*static, **static_kw))
        finally:
            for o in self.optimizers:
                del o._after_step                        # back to the class's method
        watch, keep = self._frozen_state()
        # BatchNorm buffers of nets in train() mode are moved by captured kernels (their versions change with every replay):
        # only tensors nothing in the graph writes are watched
        if any(n > 1 for n in counts.values()):
            # write_hyper() holds ONE set of scalars (learning rate, Adam bias corrections) per optimizer and replay: a second
            # captured step() of the same optimizer would be replayed with the first one's bias correction.  Not captured.
            return None
        entry = dict(graph=g, static=static, static_kw=static_kw, out=out, counts=counts, keep=keep,
                     watch=[(t, t._version) for t in watch if not self._written_in_graph(t)])
        return entry

    def _written_in_graph(self, t):
        # BatchNorm running statistics of a net that is in training mode
        for net in self.nets:
            if net.training:
                for b in net.buffers():
                    if b is t:
                        return True
        return False

    # -------------------------------------------------------------------------------------------------
    @staticmethod
    def _detached(out):
        if isinstance(out, dict):
            return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}
        if isinstance(out, (tuple, list)):
            return type(out)(v.detach() if torch.is_tensor(v) else v for v in out)
        return out.detach() if torch.is_tensor(out) else out

    def _eager(self, main, args, kw):
        self.eager_calls += 1
        out = self._detached(self.fn(*args, **kw))
        for v in (out.values() if isinstance(out, dict) else (out if isinstance(out, (tuple, list)) else (out,))):
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(main)                    # allocated on the side stream, read by the caller on its own
        return out

    def __call__(self, *args, **kw):
        self.calls += 1
        if self.enabled and _dp.exchanging(kw.get('group')):
            self.enabled, self.refused_dp = False, True       # collectives must not be captured on this stack (module docstring)
        if not self.enabled:
            self.eager_calls += 1
            return self.fn(*args, **kw)
        dev = next(a for a in args if torch.is_tensor(a) and a.is_cuda).device
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
        self._side.wait_stream(main)
        try:
            with torch.cuda.stream(self._side):
                return self._call(main, args, kw)
        finally:
            main.wait_stream(self._side)

    def _call(self, main, args, kw):
        if self.calls <= self.warmup:
            return self._eager(main, args, kw)
        key = _sig(args, kw)
        entry = self._graphs.get(key)
        if entry is not None and any(t._version != v for t, v in entry['watch']):
            entry = None                                  # a frozen weight / buffer was edited: its packed forms in the graph are stale
            del self._graphs[key]
        if entry is None:
            if len(self._graphs) >= self.max_graphs:      # e.g. the ragged last batch of an epoch: not worth a graph of its own
                return self._eager(main, args, kw)        # (optimizers in device-hyper mode refresh their scalars themselves)
            entry = self._capture(args, kw)
            if entry is None:                             # capture recorded launches only (nothing ran): run this call eagerly,
                self.enabled, self.refused_multistep = False, True      # and every later one launch by launch
                return self._eager(main, args, kw)
            self._graphs[key] = entry
        for s, a in zip(entry['static'], args):
            if torch.is_tensor(a) and s.data_ptr() != a.data_ptr():
                s.copy_(a, non_blocking=True)
        for k, a in kw.items():
            s = entry['static_kw'][k]
            if torch.is_tensor(a) and s.data_ptr() != a.data_ptr():
                s.copy_(a, non_blocking=True)
        for o in self.optimizers:
            o.write_hyper()
        entry['graph'].replay()
        self.replays += 1
        # what the skipped Python would have done: step counters, version counters of everything the update kernels wrote,
        # packed / transformed filter caches of the stepped parameters
        for o in self.optimizers:
            n = entry['counts'].get(id(o), 0)
            if n:
                o.steps += n
                torch._C._increment_version(o.params)
                ops.invalidate_packs(o.params)
        return entry['out']
