"""Data-parallel plumbing of the demo loops (SURVEY.md 8e "Partitioning"): one process per GPU, tiles sharded
across ranks, no data-path collective except the gradient exchange of ``optim`` and a handful of scalars per epoch.

 * :class:`RankStridedBatches` -- the batch sampler.  Every rank draws the SAME permutation of the tile indices
   (shared per-epoch seed), cuts it into global batches of ``world * batch_size`` and takes the elements
   ``rank, rank + world, ...`` of each.  With one rank this is exactly the reference's
   ``DataLoader(ds, batch_size, shuffle=True)`` (Demo_RSSS.py:242, Demo_USSS.py:102, Demo_WSSS.py:208): batches of
   ``batch_size`` and a shorter last one (the reference has no ``drop_last``).
 * ``ragged='weighted'`` (opt-in): no padded duplicates at all -- the last global batch is dealt out as it is, ranks hold
   ceil / floor(L / world) samples and scale their loss by ``n_local * world / L`` (``scales[i]``, ``steps._backward``), so the
   averaged gradient is exactly the gradient of the mean over the L samples (the reference's shorter last batch) -- exactly for
   everything but train-mode BatchNorm: with per-replica statistics a rank normalises over its own n_local samples (possibly 1),
   not over the L of the reference's batch (SyncBN removes that difference).  A rank left without a sample runs one flagged filler
   tile at loss scale 0: it takes part in every collective, contributes zero gradient as long as the filler's loss is finite (a
   non-finite loss times 0 is NaN and would reach every rank through the all-reduce -- the same tile would have produced it as a
   regular sample of another step), and its train-mode BatchNorm layers do see the duplicate once more in their running
   statistics (as the padded duplicates of ``ragged='pad'`` do).
 * Ragged last batch under DP: every loss is a batch mean, so mean-of-local-gradients equals the global-batch
   gradient only if all ranks hold the same number of samples.  Rule (``ragged='pad'``, default): the last global
   batch is cut into equal local batches of ``ceil(L / world)`` samples; the (< world) missing samples are taken from
   the head of the same permutation and flagged, so that epoch statistics and the confusion matrix can leave them
   out.  ``ragged='drop'`` drops the ragged global batch on every rank instead.  Either way all ranks run the same
   number of steps -- a rank that ran out of tiles early would dead-lock the next all-reduce.
 * :func:`sync_start` -- rank 0's weights, BatchNorm buffers and optimizer-visible parameters broadcast once.
 * :func:`sync_buffers` -- rank 0's BatchNorm running statistics re-broadcast before inference / checkpointing
   (per-replica statistics drift apart during training).
 * :func:`mean_scalars` / :func:`sum_counts` -- the per-epoch logging reductions (<= 9 loss averages, the 2x2
   confusion matrix: SURVEY.md 8e collective 4).
"""
import math

import torch
import torch.distributed as dist

from . import _lib


def world_info(group=None):
    """(rank, world) of this process; (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


_FORCE = {'on': bool(_lib.switch('DP_FORCE_EXCHANGE'))}


def force_exchange(on=True):
    """Run every data-parallel exchange even in a ONE-rank process group (default: a single rank skips them all).

    A 1-GPU box cannot host two RCCL ranks (RCCL refuses two ranks on one device), so without this switch the ``nccl``
    branches -- bucketed async ``all_reduce`` from the gradient-ready hooks, ``work.wait()``, the start-of-training
    broadcasts, the SyncBN sums, the epoch reductions -- would run for the first time on the 8-GPU node.  With it a
    world-size-1 ``nccl`` group exercises exactly those calls (tests/test_gpu_nccl_world1.py): the result must equal the
    no-exchange step bit for bit, since a one-rank sum is the identity and grad_scale = 1/1.  Also FCD_DP_FORCE_EXCHANGE=1."""
    prev = _FORCE['on']
    _FORCE['on'] = bool(on)
    return prev


def exchanging(group=None):
    """World size the exchange code should act on: > 1 under data parallelism, 1 when a one-rank group is forced through the
    collectives (:func:`force_exchange`), 0 when there is nothing to exchange."""
    if dist.is_available() and dist.is_initialized():
        w = dist.get_world_size(group)
        if w > 1 or _FORCE['on']:
            return w
    return 0


class RankStridedBatches:
    """Batch sampler (pass as ``DataLoader(batch_sampler=...)``).  ``pads[i]`` = number of trailing padded
    (duplicate) samples in this rank's i-th batch of the current epoch."""

    def __init__(self, n, batch_size, seed=0, shuffle=True, rank=None, world=None, ragged='pad', group=None):
        if ragged not in ('pad', 'drop', 'weighted'):
            raise ValueError("ragged must be 'pad', 'drop' or 'weighted'")
        r, w = world_info(group)
        self.n, self.batch_size, self.shuffle, self.ragged = int(n), int(batch_size), shuffle, ragged
        self.rank = r if rank is None else int(rank)
        self.world = w if world is None else int(world)
        self.seed = int(seed)
        self.pads = []
        self.scales = []      # loss scale of this rank's i-th batch (ragged='weighted': n_local * world / L for the last one)

    def set_epoch(self, seed):
        """Same value on every rank (e.g. ``base_seed * 1000 + epoch``)."""
        self.seed = int(seed)

    def _order(self):
        if not self.shuffle:
            return list(range(self.n))
        g = torch.Generator().manual_seed(self.seed)
        return torch.randperm(self.n, generator=g).tolist()

    def _plan(self):
        order, gb = self._order(), self.batch_size * self.world
        batches, pads = [], []
        full = self.n // gb
        for b in range(full):
            chunk = order[b * gb:(b + 1) * gb]
            batches.append(chunk[self.rank::self.world])
            pads.append(0)
        rest = order[full * gb:]
        scales = [1.0] * len(batches)
        if rest and self.ragged == 'weighted' and self.world > 1:
            # no duplicates: the L remaining samples are dealt round-robin, ranks end up with ceil or floor(L / world) of them
            # and weight their loss by n_local * world / L (steps._backward).  A rank left without a sample still has to take
            # part in the step's collectives: it runs ONE flagged filler sample with loss scale 0.
            mine = rest[self.rank::self.world]
            if mine:
                batches.append(mine); pads.append(0); scales.append(len(mine) * self.world / float(len(rest)))
            else:
                batches.append([order[0]]); pads.append(1); scales.append(0.0)
            self._scales = scales
            return batches, pads
        if rest and (self.ragged == 'pad' or self.world == 1):
            local = math.ceil(len(rest) / self.world)
            need = local * self.world - len(rest)
            filler = [order[i % self.n] for i in range(need)]          # head of the same permutation
            chunk = rest + filler
            mine = chunk[self.rank::self.world]
            npad = sum(1 for j in range(self.rank, len(chunk), self.world) if j >= len(rest))
            batches.append(mine)
            pads.append(npad)
            scales.append(1.0)
        self._scales = scales
        return batches, pads

    def __iter__(self):
        batches, self.pads = self._plan()
        self.scales = self._scales
        return iter(batches)

    def __len__(self):
        gb = self.batch_size * self.world
        full, rest = divmod(self.n, gb)
        return full + (1 if rest and (self.ragged in ('pad', 'weighted') or self.world == 1) else 0)


def sync_start(nets=(), optimizers=(), src=0, group=None):
    """Make every rank start from rank ``src``'s state: flat parameter buffers of the fcd optimizers, then every
    remaining parameter / buffer of ``nets`` (BatchNorm running statistics, nets without an optimizer such as a
    pre-trained Generator).  No-op on one rank."""
    if not exchanging(group):
        return
    owned = set()
    for opt in optimizers:
        opt.broadcast_state(src, group)
        owned.update(id(p) for p in opt.params)
    from . import _ops as ops
    for net in nets:
        loose = [p for p in net.parameters() if id(p) not in owned]
        for t in loose + list(net.buffers()):
            dist.broadcast(t.data, src, group=group)
        if loose:
            torch._C._increment_version(loose)
            ops.invalidate_packs(loose)


def sync_buffers(nets=(), src=0, group=None):
    """Broadcast rank ``src``'s module buffers (BatchNorm running statistics, ``num_batches_tracked``) to every rank.

    With per-replica BatchNorm the running statistics drift apart after :func:`sync_start`; eval-mode inference over
    rank-sharded tiles would then stitch a map out of ``world`` different normalisations, and the checkpoint rank 0
    writes would not reproduce it.  The demos call this before ``infer_scene`` and before every ``_save`` -- the same
    rule DistributedDataParallel applies on each forward (``broadcast_buffers``), paid once per phase instead.
    No-op on one rank."""
    if not exchanging(group):
        return
    bufs = [b for net in nets for b in net.buffers()]
    for b in bufs:
        dist.broadcast(b.data, src, group=group)
    if bufs:
        torch._C._increment_version(bufs)        # folded (conv + BN) filter caches are keyed by these versions


def mean_scalars(values, weight=1.0, group=None):
    """Weighted mean over ranks of a 1-D tensor of per-rank averages (weight = samples the rank averaged over)."""
    if not exchanging(group):
        return values
    buf = torch.cat([values.detach().double() * weight, torch.tensor([float(weight)], dtype=torch.float64,
                                                                   device=values.device)])
    dist.all_reduce(buf, group=group)
    return (buf[:-1] / buf[-1].clamp_min(1e-30)).to(values.dtype)


def sum_counts(counts, group=None):
    if exchanging(group):
        dist.all_reduce(counts, group=group)
    return counts
