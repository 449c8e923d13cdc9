"""The train-step bodies of Demo_USSS / Demo_RSSS / Demo_WSSS, data-parallel aware.

Each function is ONE iteration of the corresponding reference loop, taking the
already-constructed nets / criterion / fcd optimizers and device tensors, and
returning the loss tensors (no ``.item()`` host syncs; the caller decides when to
look).  Two modes:

 ``literal=True``  -- the reference's exact autograd call sequence, including the
    work whose results it throws away (S backward inside the D step, G / D weight
    gradients inside the S step).
 ``literal=False`` (default) -- result-identical for every parameter that is
    actually stepped, without the discarded work: the D step runs on a detached
    change map, all four D branches go through the shared net as one batch
    (``forward_pairs``), the S step freezes D and runs the eval-mode G without a
    graph.  This is the "minimal-necessary" FLOP count of SURVEY.md section 8(d).

Data parallelism (one process per GPU): gradients of the network being stepped
are all-reduced (sum) over RCCL in buckets of its flat gradient buffer, issued from
gradient-ready hooks while the backward pass is still running (``begin_overlap`` /
``allreduce_grads``, optim.py), and averaged inside the optimizer kernel; BatchNorm
uses per-replica batch statistics (DDP semantics) unless SyncBN is switched on.
"""
import contextlib

import torch
import torch.nn as nn

from . import _lib
from .Loss import region_loss


@contextlib.contextmanager
def _frozen(net):
    """Temporarily drop requires_grad on a net's parameters (data-gradient only)."""
    flags = [p.requires_grad for p in net.parameters()]
    for p in net.parameters():
        p.requires_grad_(False)
    try:
        yield
    finally:
        for p, f in zip(net.parameters(), flags):
            p.requires_grad_(f)


def _weighted(weight, term, literal):
    """``weight * term``; outside literal mode a zero weight (every demo sets ssim_weight = 0: the MS-SSIM
    value is logged, its gradient multiplied by 0 -- Demo_RSSS.py:46) drops the term's graph instead of
    back-propagating zeros through it."""
    if not literal and isinstance(weight, (int, float)) and weight == 0 and torch.is_tensor(term):
        term = term.detach()
    return weight * term


def _bcast_keep(cmask, C):
    return 1 - cmask            # (N,1,H,W) broadcasts over the C bands (reference uses .repeat)


def _masked_batch(tensors, cmask):
    """``torch.cat([t * (1 - cmask) for t in tensors], dim=0)``: every image the reference multiplies by
    ``(1 - cmask).repeat(1, C, 1, 1)`` before a Discriminator call (Demo_RSSS.py:290-300, Demo_WSSS.py:264-279), laid out as the
    batch ``Discriminator_SRGAN_simple.forward_stacked`` reads.  One HIP kernel forward, one backward (``ops.masked_stack``);
    switch FUSED_GLUE=0 runs the ATen sequence (rsub, a broadcast multiply per tensor, cat) for A/B."""
    if not _lib.switch('FUSED_GLUE'):
        keep = _bcast_keep(cmask, tensors[0].shape[1])
        return torch.cat([t * keep for t in tensors], dim=0)
    from . import _ops
    return _ops.masked_stack(tensors, cmask)


_SIDE = {}


def _side_stream(device):
    """Second HIP stream of the adversarial steps -- EXPERIMENTAL, off unless the switch STEP_OVERLAP is 1.  The Discriminator
    step -- its forward on the detached change map, backward, gradient exchange and update: many short launches on
    32x..16x16 maps that leave most of the 256 CUs idle -- is independent of the frozen-Generator forward and of the
    VGG feature passes of the Segmentor step, so it runs beside them; the Segmentor step's own Discriminator forward
    (which must see the UPDATED weights, Demo_RSSS.py:311) waits for it.  Measured on one MI355X: 98.9 -> 98.5 ms/step,
    bit-identical results -- once the packed-filter caches tell the caching allocator about
    their cross-stream readers (_ops._shared; without that the optimizer's cache invalidation let the other stream's
    allocations overwrite filters a queued kernel was still reading).  Not the default: 0.4 % is not worth a second
    stream next to the RCCL stream on the multi-GPU path, which cannot be exercised here."""
    if device.type != 'cuda' or not _lib.switch('STEP_OVERLAP'):
        return None
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        # the side stream would issue the Discriminator's collectives and read workspaces / gradient buffers the main stream
        # owns without record_stream: not supported (and not measured) under data parallelism
        raise RuntimeError('switch STEP_OVERLAP=1 is a single-GPU experiment; unset it for multi-GPU runs')
    s = _SIDE.get(device.index)
    if s is None:
        from . import _ops
        _ops.MULTI_STREAM = True            # cached packed filters are now used from two streams (see _ops._shared)
        s = _SIDE[device.index] = torch.cuda.Stream(device=device)
    return s


def _backward(loss, loss_scale=1.0, **kw):
    """``loss.backward()`` with the gradient scaled by ``loss_scale``.  Under data parallelism with
    ``dp.RankStridedBatches(ragged='weighted')`` the ranks hold DIFFERENT numbers of samples in the last batch of an epoch;
    every loss is a mean over the local batch and the exchange averages over ranks, so a rank with n_r of the L samples
    scales its loss by n_r * world / L: the averaged gradient is then the gradient of the mean over the L samples, the
    reference's shorter last batch (Demo_RSSS.py:242: no drop_last).  The reported loss values stay unscaled."""
    if loss_scale != 1.0:
        loss = loss * loss_scale
    loss.backward(**kw)


# ------------------------------------------------------------------------ RSSS
def rsss_g_pretrain_step(netG, crit, optG, x, y, region, perception_weight=0.1, ssim_weight=0, group=None, loss_scale=1.0):
    """Demo_RSSS.py:190-208."""
    optG.zero_grad()
    y_fake = netG(x)
    generator_loss, ssim_loss, perception_loss = crit(y, y_fake, region)
    g_loss = generator_loss + perception_weight * perception_loss + ssim_weight * ssim_loss
    optG.begin_overlap(group)
    _backward(g_loss, loss_scale)
    optG.allreduce_grads(group)
    optG.step()
    return dict(g_loss=g_loss, generator_loss=generator_loss, perception_loss=perception_loss, ssim_loss=ssim_loss)


def rsss_adversarial_step(netS, netD, netG, crit, optS, optD, x, y, region, perception_weight=0.1,
                          ssim_weight=0, l1_weight=0.02, g_weight=0.5, d_weight=1, r_weight=2,
                          discriminator_continuous=True, literal=False, group=None, loss_scale=1.0, d_share=None):
    """Demo_RSSS.py:285-332 (netG in eval mode, Demo_RSSS.py:240).  ``d_share``: the Discriminator step sends the shared masked x
    through D's net once (True) or twice as the reference does (False); None follows the switch D_SHARE (default 1).  A keyword of
    the step, not an environment read inside it."""
    cmap = netS(x, y)
    cmask = cmap if discriminator_continuous else (torch.sign(cmap - 0.5) + 1) / 2
    y_unc = y * (1 - region) + x * region
    # ---- D step
    if literal:
        keep = _bcast_keep(cmask, x.shape[1])
        x_mask, y_mask = x * keep, y * keep
        c_out = netD(x_mask, y_mask)
        nc_out = netD(x * keep, y_unc * keep)
        optD.zero_grad()
        d_loss = 1 + nc_out.mean() - c_out.mean()
        optD.begin_overlap(group)
        _backward(d_loss, loss_scale, retain_graph=True)
    else:
        side = _side_stream(x.device)
        main = torch.cuda.current_stream(x.device) if side is not None else None
        if side is not None:
            side.wait_stream(main)              # cmap, y_unc are ready
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            # x_unc == x (Demo_RSSS.py:297): both calls take the same masked x, which goes through D's net once
            # (``d_share=False`` / switch D_SHARE=0: twice, as a batch of four groups -- the round-4 form, for A/B runs)
            if not (_lib.switch('D_SHARE') if d_share is None else d_share):
                c_out, nc_out = netD.forward_stacked(_masked_batch([x, y, x, y_unc], cmask.detach()), 2)
            else:
                c_out, nc_out = netD.forward_shared_first(_masked_batch([x, y, y_unc], cmask.detach()), 2)
            optD.zero_grad()
            d_loss = 1 + nc_out.mean() - c_out.mean()
            optD.begin_overlap(group)
            _backward(d_loss, loss_scale)
    if literal or side is None:
        optD.allreduce_grads(group)
        optD.step()
    else:
        with torch.cuda.stream(side):
            optD.allreduce_grads(group)
            optD.step()
            d_loss.record_stream(main)
    # ---- S step
    if literal:
        c_out = netD(x_mask, y_mask)
        y_fake = netG(x)
        generator_loss, ssim_loss, perception_loss = crit(y, y_fake, cmap)
    else:
        with torch.no_grad():
            y_fake = netG(x)
        generator_loss, ssim_loss, perception_loss = crit(y, y_fake, cmap)
        if side is not None:
            main.wait_stream(side)              # D's updated weights and running statistics
        with _frozen(netD):
            c_out = netD.forward_stacked(_masked_batch([x, y], cmask), 1)[0]
    g_loss = generator_loss + perception_weight * perception_loss + _weighted(ssim_weight, ssim_loss, literal)
    l1_loss = region_loss(cmap, region, 'l1')
    s_d_loss = c_out.mean()
    r_loss = region_loss(cmap, 1 - region, 'mse')
    s_loss = d_weight * s_d_loss + l1_weight * l1_loss + g_weight * g_loss + r_weight * r_loss
    optS.zero_grad()
    optS.begin_overlap(group)
    _backward(s_loss, loss_scale)
    optS.allreduce_grads(group)
    optS.step()
    return dict(d_loss=d_loss, s_loss=s_loss, s_d_loss=s_d_loss, g_loss=g_loss, l1_loss=l1_loss, r_loss=r_loss,
                generator_loss=generator_loss, ssim_loss=ssim_loss, perception_loss=perception_loss, cmap=cmap)


# ------------------------------------------------------------------------ WSSS
def wsss_adversarial_step(netS, netD, netG, crit, optS, optD, x, y, x_nc, y_nc, perception_weight=0.5,
                          ssim_weight=0, g_weight=0.2, l1_weight=1.6, d_weight=1, nc_weight=1.5,
                          discriminator_continuous=True, literal=False, group=None, loss_scale=1.0):
    """Demo_WSSS.py:249-323 (netG in eval mode, Demo_WSSS.py:206).  The unchanged pair
    is masked with the CHANGED pair's map (:278-279)."""
    cmap = netS(x, y)
    cmask = cmap if discriminator_continuous else (torch.sign(cmap - 0.5) + 1) / 2
    if literal:
        keep = _bcast_keep(cmask, x.shape[1])
        x_mask, y_mask = x * keep, y * keep
        c_out = netD(x_mask, y_mask)
        ncmap = netS(x_nc, y_nc)
        nc_out = netD(x_nc * keep, y_nc * keep)
        optD.zero_grad()
        d_loss = 1 + nc_out.mean() - c_out.mean()
        optD.begin_overlap(group)
        _backward(d_loss, loss_scale, retain_graph=True)
    else:
        ncmap = netS(x_nc, y_nc)
        c_out, nc_out = netD.forward_stacked(_masked_batch([x, y, x_nc, y_nc], cmask.detach()), 2)
        optD.zero_grad()
        d_loss = 1 + nc_out.mean() - c_out.mean()
        optD.begin_overlap(group)
        _backward(d_loss, loss_scale)
    optD.allreduce_grads(group)
    optD.step()
    nc_loss = torch.mean(torch.pow(ncmap, 2))
    if literal:
        c_out = netD(x_mask, y_mask)
        if g_weight != 0:
            y_fake = netG(x)
    else:
        with _frozen(netD):
            c_out = netD.forward_stacked(_masked_batch([x, y], cmask), 1)[0]
        if g_weight != 0:
            with torch.no_grad():
                y_fake = netG(x)
    if g_weight != 0:
        generator_loss, ssim_loss, perception_loss = crit(y, y_fake, cmap)
    else:
        generator_loss = ssim_loss = perception_loss = torch.zeros((), device=x.device)
    g_loss = generator_loss + perception_weight * perception_loss + _weighted(ssim_weight, ssim_loss, literal)
    l1_loss = torch.mean(abs(cmap))
    s_d_loss = c_out.mean()
    s_loss = d_weight * s_d_loss + l1_weight * l1_loss + g_weight * g_loss + nc_weight * nc_loss
    optS.zero_grad()
    optS.begin_overlap(group)
    _backward(s_loss, loss_scale)
    optS.allreduce_grads(group)
    optS.step()
    return dict(d_loss=d_loss, s_loss=s_loss, s_d_loss=s_d_loss, g_loss=g_loss, l1_loss=l1_loss, nc_loss=nc_loss,
                generator_loss=generator_loss, ssim_loss=ssim_loss, perception_loss=perception_loss, cmap=cmap,
                ncmap=ncmap)


# ------------------------------------------------------------------------ USSS
def usss_g_pretrain_step(netG, crit, optG, x, y, perception_weight=0.4, ssim_weight=0, group=None, loss_scale=1.0, literal=False):
    """Demo_USSS.py:142-159 (cmap = 0)."""
    optG.zero_grad()
    y_fake = netG(x)
    cmap = torch.zeros((x.shape[0], 1, x.shape[2], x.shape[3]), device=x.device)
    generator_loss, l1_loss, perception_loss, ssim_loss = crit(y, y_fake, cmap)
    loss = generator_loss + perception_weight * perception_loss + _weighted(ssim_weight, ssim_loss, literal)
    optG.begin_overlap(group)
    _backward(loss, loss_scale)
    optG.allreduce_grads(group)
    optG.step()
    return dict(loss=loss, generator_loss=generator_loss, perception_loss=perception_loss, ssim_loss=ssim_loss)


def usss_s_pretrain_step(netS, netG, crit, optS, x, y, perception_weight=0.4, l1_weight=0.65, ssim_weight=0,
                         literal=False, group=None, loss_scale=1.0):
    """Demo_USSS.py:219-228: G forward in train mode (its BN running stats keep moving),
    only S is stepped."""
    if literal:
        y_fake = netG(x)
    else:
        with torch.no_grad():
            y_fake = netG(x)
    cmap = netS(x, y)
    generator_loss, l1_loss, perception_loss, ssim_loss = crit(y, y_fake, cmap)
    net_loss = generator_loss + l1_weight * l1_loss + perception_weight * perception_loss + _weighted(ssim_weight, ssim_loss, literal)
    optS.zero_grad()
    optS.begin_overlap(group)
    _backward(net_loss, loss_scale)
    optS.allreduce_grads(group)
    optS.step()
    return dict(net_loss=net_loss, generator_loss=generator_loss, l1_loss=l1_loss, perception_loss=perception_loss,
                ssim_loss=ssim_loss, cmap=cmap)


def usss_joint_step(netS, netG, crit, optS, optG, x, y, perception_weight=0.4, l1_weight=0.65, ssim_weight=0,
                    literal=False, group=None, loss_scale=1.0):
    """Demo_USSS.py:310-341.  The reference backpropagates ``Loss`` (retain_graph) and then
    ``NetLoss = Loss + l1_weight*l1`` over the same graph: G ends with grad(Loss)+grad(NetLoss)
    = 2*grad(Loss) (l1 does not depend on G), S with grad(NetLoss) only (zero_grad in
    between).  literal=False does ONE backward of NetLoss and doubles G's gradient."""
    optG.zero_grad()
    y_fake = netG(x)
    cmap = netS(x, y)
    generator_loss, l1_loss, perception_loss, ssim_loss = crit(y, y_fake, cmap)
    loss = generator_loss + perception_weight * perception_loss + _weighted(ssim_weight, ssim_loss, literal)
    net_loss = generator_loss + l1_weight * l1_loss + perception_weight * perception_loss + _weighted(ssim_weight, ssim_loss, literal)
    if literal:
        _backward(loss, loss_scale, retain_graph=True)
        optS.zero_grad()
        optS.begin_overlap(group)          # G collects gradients from BOTH backward passes: exchanged afterwards
        _backward(net_loss, loss_scale)
        optG.allreduce_grads(group)
    else:
        optS.zero_grad()
        optG.begin_overlap(group)
        optS.begin_overlap(group)
        _backward(net_loss, loss_scale)
        optG.allreduce_grads(group)         # (waits for G's buckets before the in-place doubling)
        optG.flat_g.mul_(2.0)
    optS.allreduce_grads(group)
    optG.step()
    optS.step()
    return dict(loss=loss, net_loss=net_loss, generator_loss=generator_loss, l1_loss=l1_loss,
                perception_loss=perception_loss, ssim_loss=ssim_loss, cmap=cmap)


# --------------------------------------------------------------------- inference
@torch.no_grad()
def infer_density(netS, x, y, prob_thresh=0.5):
    """Inference body of the demos (Demo_RSSS.py:457-491, Demo_USSS.py:413-470): change-density
    map and thresholded binary map.  ``netS.eval()`` must have been called; BatchNorm is then
    folded into the convolutions (Module.DoubleConv).  Demo_WSSS keeps train() mode on purpose
    (Demo_WSSS.py:389-391) -- in that case the regular batch-statistics path runs."""
    cmap = netS(x, y)
    return cmap, (cmap > prob_thresh)


@torch.no_grad()
def infer_density_raw(netS, x_raw, y_raw, valid, stats, prob_thresh=0.5):
    """``infer_density`` on raw tiles: NORMALIZE (CommonFunc.py:199-224) folded into the first convolution
    (``Segmentor.forward_raw``) -- no normalisation pass on the host or the device."""
    cmap = netS.forward_raw(x_raw, y_raw, valid, stats)
    return cmap, (cmap > prob_thresh)


# ------------------------------------------------------- on-device confusion matrix
def confusion_counts(cmap, ref_changed, prob_thresh=0.5, group=None):
    """2x2 confusion counts of the thresholded map vs. a {0,1} reference on device
    (replaces the per-sample D2H + NumPy loop of Demo_RSSS.py:345-354); returns an
    int64 tensor [tn, fp, fn, tp], all-reduced across ranks when distributed."""
    from . import dp
    pred = cmap > prob_thresh
    ref = ref_changed > 0.5
    counts = torch.stack([(~pred & ~ref).sum(), (pred & ~ref).sum(), (~pred & ref).sum(), (pred & ref).sum()])
    return dp.sum_counts(counts, group)
