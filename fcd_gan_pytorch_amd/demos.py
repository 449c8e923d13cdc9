"""Runnable restatements of the reference's three demo scripts on the HIP path
(Demo_USSS.py / Demo_RSSS.py / Demo_WSSS.py are ``if __name__ == '__main__'`` blocks with
hard-coded paths; these are the same phases as functions with the knobs as arguments).

Only the orchestration lives here: tile feeding (``tiles``), the step bodies (``steps``),
the LR schedule, on-device metrics (``metrics``), inference + centre write-back.  Logging
to TensorBoard, ``Para*.txt`` dumps and ETA printing are out of scope.
"""
import os

import numpy as np
import torch
from torch.utils.data import DataLoader

from . import Loss, Module, dp, metrics, optim, steps, tiles


class _Epoch:
    """One pass over a dataset, data-parallel aware: rank-strided batches off a shared per-epoch permutation
    (``dp.RankStridedBatches``), pinned-memory prefetch, and per-batch sample weights that leave out the padded
    duplicates of a ragged last batch.  Iterating yields ``(batch, n_real, weight)`` with
    ``weight = n_real / len(dataset)`` so that ``sum(weight * batch_mean)`` over all ranks is the epoch mean the
    reference logs (Demo_RSSS.py:334-343)."""

    def __init__(self, ds, batch_size, seed, device, shuffle=True, wrap=None, ragged='pad'):
        self.ds, self.device, self.wrap = ds, device, wrap
        self.sampler = dp.RankStridedBatches(len(ds), batch_size, seed=seed, shuffle=shuffle, ragged=ragged)

    def __iter__(self):
        loader = DataLoader(self.ds, batch_sampler=self.sampler)
        it = self.wrap(loader) if self.wrap else loader
        for i, batch in enumerate(tiles.Prefetcher(it, self.device)):
            n = batch[0].shape[0]
            real = n - self.sampler.pads[i]
            w = _Weight(real / float(len(self.ds)))
            w.loss_scale = self.sampler.scales[i]         # != 1 only in the last batch under ragged='weighted' (pass to the step)
            yield batch, real, w


class _Weight(float):
    """The batch's epoch weight (a float) that also carries the loss scale of the step (``dp`` ragged='weighted')."""
    loss_scale = 1.0


def _real_mask(like, real):
    """(N,1,1,1) bool: True for the first ``real`` samples of the batch (padded duplicates come last)."""
    m = torch.zeros((like.shape[0], 1, 1, 1), dtype=torch.bool, device=like.device)
    m[:real] = True
    return m


def _epoch_mean(total):
    """Sum of the ranks' weighted partial sums (each rank saw its share of the samples)."""
    return dp.sum_counts(total.double())


def _save(net, path):
    """``torch.save(net.state_dict(), path)`` on rank 0 (Demo_RSSS.py:507-514, Demo_USSS.py:477-481)."""
    if path and dp.world_info()[0] == 0:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        torch.save(net.state_dict(), path)


def _tile_stats(scene_x, scene_y, patch_size, stats_txt=None):
    """Dataset_meanstd over the NON-overlapped tiling of the scene pair (Demo_USSS.py:88-95): per-tile means weighted
    by valid-pixel counts, cached in the reference's two text files when ``stats_txt=(path_x, path_y)``."""
    ds = tiles.PairTileDataset(scene_x, scene_y, None, patch_size, (0, 0))
    if stats_txt:
        return tuple(np.asarray(v, np.float64) for v in tiles.dataset_meanstd(stats_txt[0], stats_txt[1], ds))
    mx, my = tiles.dataset_mean(ds)
    sx, sy = tiles.dataset_std(ds, mx, my)
    return tuple(v.double().numpy() for v in (mx, sx, my, sy))


def demo_usss(scene_x, scene_y, ref=None, device='cuda', patch_size=(220, 220), overlap_padding=(10, 10),
              epochs_g=50, epochs_s=50, epochs_joint=100, batch_size=10, learning_rate=2e-4,
              perception_weight=0.4, l1_weight=0.65, ssim_weight=0, perception_perBand=True,
              prob_thresh=0.5, gt_map=(1, 2), pre_map=(0, 1), seed=0, out_density=None, out_color=None,
              log=None, allow_seeded=False, stats_txt=None, save_s=None, save_g=None, ragged='pad'):
    """Demo_USSS.py:29-501: G pre-train -> S pre-train -> joint training -> inference with centre write-back, with the
    reference's LR schedules (Demo_USSS.py:133,201,298-299) and per-tile dataset statistics (:88-95).
    ``scene_x/scene_y/ref``: (bands,H,W) arrays or TIFF paths; ``save_s`` / ``save_g``: ``.pkl`` paths
    (Demo_USSS.py:477-481).  ``batch_size`` is per rank under data parallelism.
    Returns dict(netS, netG, density (1,H,W) float32, color (1,H,W), evaluator, history, vgg_pretrained)."""
    dev = torch.device(device)
    if isinstance(scene_x, str):
        scene_x = tiles.read_tiff(scene_x)
    if isinstance(scene_y, str):
        scene_y = tiles.read_tiff(scene_y)
    if isinstance(ref, str):
        ref = tiles.read_tiff(ref)
    stats = _tile_stats(scene_x, scene_y, patch_size, stats_txt)
    ds = tiles.PairTileDataset(scene_x, scene_y, ref, patch_size, overlap_padding, stats=stats)
    nband = scene_x.shape[0]
    torch.manual_seed(seed)
    netS = Module.Segmentor(n_channels=nband, bilinear=True).to(dev)
    netG = Module.Generator(n_channels=nband).to(dev)
    crit = Loss.CNetLoss(channel=nband, perception_layer=1, perception_perBand=perception_perBand,
                         allow_seeded=allow_seeded).to(dev)
    netS.train(); netG.train()                                                    # Demo_USSS.py:116-117
    optS = optim.Adam(netS.parameters(), lr=learning_rate, betas=(0.9, 0.99))
    optG = optim.Adam(netG.parameters(), lr=learning_rate, betas=(0.9, 0.99))
    dp.sync_start((netS, netG), (optS, optG))
    hist = {'g': [], 's': [], 'joint': []}
    acc = metrics.Evaluator(2)
    kw = dict(perception_weight=perception_weight, ssim_weight=ssim_weight)

    for ep in range(epochs_g):                                                   # Demo_USSS.py:126-189
        optim.adjust_learning_rate(optG, ep, lr_start=1e-5, lr_max=3e-4, lr_warm_up_epoch=10, lr_sustain_epochs=10)
        tot = torch.zeros((), device=dev)
        for (x, y, item, r), real, w in _Epoch(ds, batch_size, seed * 1000 + ep, dev, ragged=ragged):
            out = steps.usss_g_pretrain_step(netG, crit, optG, x, y, **kw, loss_scale=w.loss_scale)
            tot += out['loss'].detach() * w
        hist['g'].append(float(_epoch_mean(tot)))
        if log:
            log('G pre-train epoch %d loss %.4f' % (ep + 1, hist['g'][-1]))
    for ep in range(epochs_s):                                                   # Demo_USSS.py:194-286
        optim.adjust_learning_rate(optS, ep, lr_start=1e-5, lr_max=3e-4, lr_warm_up_epoch=10, lr_sustain_epochs=10)
        tot = torch.zeros((), device=dev)
        for (x, y, item, r), real, w in _Epoch(ds, batch_size, seed * 1000 + 1000 + ep, dev, ragged=ragged):
            out = steps.usss_s_pretrain_step(netS, netG, crit, optS, x, y, l1_weight=l1_weight, **kw, loss_scale=w.loss_scale)
            tot += out['net_loss'].detach() * w
        hist['s'].append(float(_epoch_mean(tot)))
        if log:
            log('S pre-train epoch %d loss %.4f' % (ep + 1, hist['s'][-1]))
    for ep in range(epochs_joint):                                               # Demo_USSS.py:291-400
        optim.adjust_learning_rate(optS, ep, lr_start=1e-5, lr_max=1e-4)
        optim.adjust_learning_rate(optG, ep, lr_start=1e-5, lr_max=1e-4)
        tot = torch.zeros((), device=dev)
        acc.reset()
        for (x, y, item, r), real, w in _Epoch(ds, batch_size, seed * 1000 + 2000 + ep, dev, ragged=ragged):
            out = steps.usss_joint_step(netS, netG, crit, optS, optG, x, y, l1_weight=l1_weight, **kw, loss_scale=w.loss_scale)
            tot += out['net_loss'].detach() * w
            if ref is not None:
                acc.add_batch_map(r, metrics.threshold_map(out['cmap'].detach(), prob_thresh), gt_map, pre_map,
                                  valid=_real_mask(r, real).expand_as(r))
        hist['joint'].append(float(_epoch_mean(tot)))
        if log:
            log('joint epoch %d loss %.4f' % (ep + 1, hist['joint'][-1]))

    dp.sync_buffers((netS, netG))            # one set of BatchNorm statistics for the stitched map AND the checkpoint
    res = infer_scene(netS, ds, dev, batch_size=batch_size, prob_thresh=prob_thresh, gt_map=gt_map, pre_map=pre_map)
    if out_density and dp.world_info()[0] == 0:
        tiles.write_tiff(out_density, res['density'])
    if out_color and dp.world_info()[0] == 0:
        tiles.write_tiff(out_color, res['color'])
    _save(netS, save_s)
    _save(netG, save_g)
    res.update(netS=netS, netG=netG, history=hist, train_evaluator=acc, vgg_pretrained=crit.loss_perception.pretrained)
    return res


@torch.no_grad()
def infer_scene(netS, ds, device, batch_size=10, prob_thresh=0.5, gt_map=(1, 2), pre_map=(0, 1), eval_mode=True):
    """Inference over every tile of a PairTileDataset with centre write-back
    (Demo_USSS.py:404-470, Demo_RSSS.py:451-491): density map (float32), TP/FP/FN colour codes
    and the evaluator over the owned centres.  ``eval_mode=False`` keeps train() statistics
    like Demo_WSSS.py:389-391.  Under data parallelism the tiles are sharded rank-strided; every
    centre is owned by exactly one tile, so the stitched maps are the SUM over ranks; in eval mode rank 0's BatchNorm
    running statistics are broadcast first (``dp.sync_buffers``) so every tile is normalised alike and the map equals
    what rank 0's checkpoint reproduces."""
    dev = torch.device(device)
    if eval_mode:
        dp.sync_buffers((netS,))
    was_training = netS.training
    netS.train(not eval_mode)
    grid = ds.grid
    density = np.zeros((1, grid.ysize, grid.xsize), np.float32)
    color = np.zeros((1, grid.ysize, grid.xsize), np.float32)
    acc = metrics.Evaluator(2)
    for (x, y, item, r), real, _ in _Epoch(ds, batch_size, 0, dev, shuffle=False):
        cmap, mask = steps.infer_density(netS, x, y, prob_thresh)
        maskf = mask.to(cmap.dtype)
        valid = torch.zeros_like(mask)
        items = item.tolist()[:real]                       # padded duplicates belong to another rank
        for i, it in enumerate(items):
            r0, r1, c0, c1 = grid.eff_range(it)
            valid[i, :, r0:r1, c0:c1] = True
        if ds.ref is not None:
            acc.add_batch_map(r, maskf, gt_map, pre_map, valid=valid)
            codes = metrics.changemap_codes(maskf, r, True, ref_map=gt_map, dt_map=pre_map)
        else:
            codes = metrics.changemap_codes(maskf, maskf, False, dt_map=pre_map)
        cm_h, co_h = cmap.cpu().numpy(), codes.cpu().numpy()
        for i, it in enumerate(items):
            grid.write_center(density, cm_h[i], it)
            grid.write_center(color, co_h[i], it)
    if dp.exchanging():
        both = torch.from_numpy(np.stack([density, color])).to(dev)
        dp.sum_counts(both)
        density, color = (a.copy() for a in both.cpu().numpy())
    netS.train(was_training)
    return dict(density=density, color=color, evaluator=acc)


def _valid_centres(dataset, like, items, real):
    valid = torch.zeros_like(like, dtype=torch.bool)
    for i, it in enumerate(items[:real]):
        r0, r1, c0, c1 = dataset.eff_range(it) if hasattr(dataset, 'eff_range') else dataset.grid.eff_range(it)
        valid[i, :, r0:r1, c0:c1] = True
    return valid


def demo_rsss(dataset, device='cuda', n_channels=4, epochs_g=50, epochs_adv=100, init_batch_size=20, batch_size=12,
              learning_rate=5e-5, perception_weight=0.1, ssim_weight=0, perception_perBand=True, l1_weight=0.02,
              g_weight=0.5, d_weight=1, r_weight=2, prob_thresh=0.5, gt_map=(1, 2), pre_map=(0, 1), seed=0,
              netG_state=None, log=None, allow_seeded=False, load_g=None, save_s=None, save_g=None, save_d=None, ragged='pad'):
    """Demo_RSSS.py:27-538 on a ``datasets.MultiSceneDataset`` / ``RegionTileDataset`` (tuples
    ``(x, y, item, ref, region)``): G pre-training on the region-masked reconstruction (skipped when a
    generator checkpoint is given -- ``load_g``: path of a ``GModel.pkl`` as the reference writes it,
    Demo_RSSS.py:167-171, or ``netG_state``: a state_dict), netG.eval(), adversarial D/S loop with the
    reference's LR schedules, per-epoch on-device accuracy over the owned tile centres, ``.pkl`` checkpoints
    (``save_s/save_g/save_d``, Demo_RSSS.py:507-514).  Batch sizes are per rank under data parallelism."""
    dev = torch.device(device)
    torch.manual_seed(seed)
    netD = Module.Discriminator_SRGAN_simple(n_channels=n_channels).to(dev)
    netS = Module.Segmentor(n_channels=n_channels, bilinear=True).to(dev)
    netG = Module.Generator(n_channels=n_channels).to(dev)
    netS.train(); netG.train(); netD.train()
    crit = Loss.CGeneratorLoss(channel=n_channels, perception_layer=1, perception_perBand=perception_perBand,
                               allow_seeded=allow_seeded).to(dev)
    optG = optim.Adam(netG.parameters(), lr=learning_rate, betas=(0.9, 0.99))
    optS = optim.RMSprop(netS.parameters(), lr=learning_rate)
    optD = optim.RMSprop(netD.parameters(), lr=learning_rate)
    hist = {'g': [], 'adv': []}
    if load_g is not None and os.path.exists(load_g):                            # Demo_RSSS.py:167-171
        netG_state = torch.load(load_g, map_location=dev)
    if netG_state is not None:
        netG.load_state_dict(netG_state)
        epochs_g = 0
    dp.sync_start((netS, netD, netG), (optS, optD, optG))
    for ep in range(epochs_g):                                                   # Demo_RSSS.py:175-236
        optim.adjust_learning_rate(optG, ep, lr_start=1e-5, lr_max=3e-4, lr_warm_up_epoch=10, lr_sustain_epochs=10)
        tot = torch.zeros((), device=dev)
        for (x, y, item, r, region), real, w in _Epoch(dataset, init_batch_size, seed * 1000 + ep, dev, ragged=ragged):
            out = steps.rsss_g_pretrain_step(netG, crit, optG, x, y, region, perception_weight, ssim_weight, loss_scale=w.loss_scale)
            tot += out['g_loss'].detach() * w
        hist['g'].append(float(_epoch_mean(tot)))
        if log:
            log('G pre-train epoch %d g_loss %.4f' % (ep + 1, hist['g'][-1]))
    netG.eval()                                                                   # Demo_RSSS.py:240
    acc = metrics.Evaluator(2)
    for ep in range(epochs_adv):                                                  # Demo_RSSS.py:246-396
        optim.adjust_learning_rate(optS, ep, lr_start=1e-4, lr_max=1e-3, lr_warm_up_epoch=5)
        optim.adjust_learning_rate(optD, ep, lr_start=5e-6, lr_max=5e-5, lr_min=5e-7, lr_warm_up_epoch=5)
        acc.reset()
        sums = torch.zeros(3, device=dev)
        for (x, y, item, r, region), real, w in _Epoch(dataset, batch_size, seed * 1000 + 500 + ep, dev, ragged=ragged):
            out = steps.rsss_adversarial_step(netS, netD, netG, crit, optS, optD, x, y, region,
                                              perception_weight=perception_weight, ssim_weight=ssim_weight,
                                              l1_weight=l1_weight, g_weight=g_weight, d_weight=d_weight,
                                              r_weight=r_weight, loss_scale=w.loss_scale)
            sums += torch.stack([out['d_loss'].detach(), out['s_loss'].detach(), out['g_loss'].detach()]) * w
            valid = _valid_centres(dataset, out['cmap'], item.tolist(), real)
            acc.add_batch_map(r, metrics.threshold_map(out['cmap'].detach(), prob_thresh), gt_map, pre_map, valid=valid)
        hist['adv'].append([float(v) for v in _epoch_mean(sums)] + [float(acc.Pixel_F1_score())])
        if log:
            log('adv epoch %d d %.4f s %.4f g %.4f F1 %.4f' % ((ep + 1,) + tuple(hist['adv'][-1])))
    dp.sync_buffers((netS, netG, netD))      # the saved .pkl = the statistics every rank infers with from here on
    _save(netS, save_s)
    _save(netG, save_g)
    _save(netD, save_d)
    return dict(netS=netS, netD=netD, netG=netG, history=hist, evaluator=acc,
                vgg_pretrained=crit.loss_perception.pretrained)


def demo_wsss(changed_ds, unchanged_ds, device='cuda', n_channels=3, epochs_g=50, epochs_adv=50, unc_batch_size=50,
              batch_size=15, perception_weight=0.5, ssim_weight=0, g_weight=0.2, l1_weight=1.6, d_weight=1,
              nc_weight=1.5, seed=0, netG_state=None, log=None, allow_seeded=False, load_g=None, save_s=None,
              save_g=None, save_d=None, ragged='pad'):
    """Demo_WSSS.py:27-483 on datasets of (x, y, ...) tuples: G pre-training on UNCHANGED pairs
    with cmap = 0 (Demo_WSSS.py:152-176; skipped with ``load_g`` / ``netG_state``, :131-135), netG.eval(),
    adversarial loop over (changed, unchanged) pairs re-matched every epoch (``PairingDataset.order_reset``,
    same seed on every rank), ``.pkl`` checkpoints (:454-461)."""
    from .datasets import PairingDataset
    dev = torch.device(device)
    torch.manual_seed(seed)
    netD = Module.Discriminator_SRGAN_simple(n_channels).to(dev)
    netS = Module.Segmentor(n_channels=n_channels, bilinear=True).to(dev)
    netG = Module.Generator(n_channels=n_channels).to(dev)
    netS.train(); netD.train(); netG.train()
    crit = Loss.CGeneratorLoss(channel=n_channels, perception_layer=1, perception_perBand=False,
                               allow_seeded=allow_seeded).to(dev)
    optG = optim.Adam(netG.parameters(), lr=5e-4, betas=(0.9, 0.99))
    optS = optim.RMSprop(netS.parameters(), lr=1e-3)
    optD = optim.RMSprop(netD.parameters(), lr=1e-5)
    hist = {'g': [], 'adv': []}
    if load_g is not None and os.path.exists(load_g):                            # Demo_WSSS.py:131-135
        netG_state = torch.load(load_g, map_location=dev)
    if netG_state is not None:
        netG.load_state_dict(netG_state)
        epochs_g = 0
    if g_weight == 0:
        epochs_g = 0
    dp.sync_start((netS, netD, netG), (optS, optD, optG))
    for ep in range(epochs_g):
        optim.adjust_learning_rate(optG, ep, lr_start=1e-5, lr_max=3e-4, lr_warm_up_epoch=10, lr_sustain_epochs=10)
        tot = torch.zeros((), device=dev)
        for batch, real, w in _Epoch(unchanged_ds, unc_batch_size, seed * 1000 + ep, dev, ragged=ragged):
            x, y = batch[0], batch[1]
            optG.zero_grad()
            y_fake = netG(x)
            cmap = torch.zeros((x.shape[0], 1, x.shape[2], x.shape[3]), device=dev)
            gen, ssim, perc = crit(y, y_fake, cmap)
            g_loss = gen + perception_weight * perc + ssim_weight * ssim
            optG.begin_overlap()
            steps._backward(g_loss, w.loss_scale)
            optG.allreduce_grads()
            optG.step()
            tot += g_loss.detach() * w
        hist['g'].append(float(_epoch_mean(tot)))
        if log:
            log('G pre-train epoch %d g_loss %.4f' % (ep + 1, hist['g'][-1]))
    netG.eval()                                                                   # Demo_WSSS.py:206
    pairs = PairingDataset(changed_ds, unchanged_ds, random_assign=False, seed=seed)

    def flat(loader):
        for cds, ncds in loader:
            yield (cds[0], cds[1], ncds[0], ncds[1])
    for ep in range(epochs_adv):                                                  # Demo_WSSS.py:209-323
        optim.adjust_learning_rate(optS, ep, lr_start=1e-4, lr_max=1e-3, lr_warm_up_epoch=5)
        optim.adjust_learning_rate(optD, ep, lr_start=1e-6, lr_max=1e-5, lr_min=1e-8, lr_warm_up_epoch=5)
        pairs.order_reset(seed=seed * 7919 + ep)                                   # same pairing on every rank
        sums = torch.zeros(2, device=dev)
        for (x, y, x_nc, y_nc), real, w in _Epoch(pairs, batch_size, seed * 1000 + 700 + ep, dev, wrap=flat, ragged=ragged):
            out = steps.wsss_adversarial_step(netS, netD, netG, crit, optS, optD, x, y, x_nc, y_nc,
                                              perception_weight=perception_weight, ssim_weight=ssim_weight,
                                              g_weight=g_weight, l1_weight=l1_weight, d_weight=d_weight,
                                              nc_weight=nc_weight, loss_scale=w.loss_scale)
            sums += torch.stack([out['d_loss'].detach(), out['s_loss'].detach()]) * w
        hist['adv'].append([float(v) for v in _epoch_mean(sums)])
        if log:
            log('adv epoch %d d %.4f s %.4f' % ((ep + 1,) + tuple(hist['adv'][-1])))
    dp.sync_buffers((netS, netG, netD))
    _save(netS, save_s)
    _save(netG, save_g)
    _save(netD, save_d)
    return dict(netS=netS, netD=netD, netG=netG, history=hist, vgg_pretrained=crit.loss_perception.pretrained)
