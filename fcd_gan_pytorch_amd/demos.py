"""Runnable restatements of the reference's three demo scripts on the HIP path
(Demo_USSS.py / Demo_RSSS.py / Demo_WSSS.py are ``if __name__ == '__main__'`` blocks with
hard-coded paths; these are the same phases as functions with the knobs as arguments).

Only the orchestration lives here: tile feeding (``tiles``), the step bodies (``steps``),
the LR schedule, on-device metrics (``metrics``), inference + centre write-back.  Logging
to TensorBoard, ``Para*.txt`` dumps and ETA printing are out of scope.
"""
import warnings

import numpy as np
import torch
from torch.utils.data import DataLoader

from . import Loss, Module, metrics, optim, steps, tiles


def _loader(ds, batch_size, shuffle, seed):
    g = torch.Generator().manual_seed(seed)
    return DataLoader(ds, batch_size=batch_size, shuffle=shuffle, generator=g)


def _stats(scene):
    flat = scene.reshape(scene.shape[0], -1).astype(np.float64)
    return flat.mean(1), flat.std(1)


def demo_usss(scene_x, scene_y, ref=None, device='cuda', patch_size=(220, 220), overlap_padding=(10, 10),
              epochs_g=50, epochs_s=50, epochs_joint=100, batch_size=10, learning_rate=2e-4,
              perception_weight=0.4, l1_weight=0.65, ssim_weight=0, perception_perBand=True,
              prob_thresh=0.5, gt_map=(1, 2), pre_map=(0, 1), seed=0, out_density=None, out_color=None,
              log=None):
    """Demo_USSS.py:29-501: G pre-train -> S pre-train -> joint training -> inference with
    centre write-back.  ``scene_x/scene_y/ref``: (bands,H,W) arrays or TIFF paths.
    Returns dict(netS, netG, density (1,H,W) float32, color (1,H,W), evaluator, history)."""
    dev = torch.device(device)
    if isinstance(scene_x, str):
        scene_x = tiles.read_tiff(scene_x)
    if isinstance(scene_y, str):
        scene_y = tiles.read_tiff(scene_y)
    if isinstance(ref, str):
        ref = tiles.read_tiff(ref)
    mx, sx = _stats(scene_x)
    my, sy = _stats(scene_y)
    ds = tiles.PairTileDataset(scene_x, scene_y, ref, patch_size, overlap_padding, stats=(mx, sx, my, sy))
    nband = scene_x.shape[0]
    torch.manual_seed(seed)
    netS = Module.Segmentor(n_channels=nband, bilinear=True).to(dev)
    netG = Module.Generator(n_channels=nband).to(dev)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        crit = Loss.CNetLoss(channel=nband, perception_layer=1, perception_perBand=perception_perBand).to(dev)
    netS.train(); netG.train()                                                    # Demo_USSS.py:116-117
    optS = optim.Adam(netS.parameters(), lr=learning_rate, betas=(0.9, 0.99))
    optG = optim.Adam(netG.parameters(), lr=learning_rate, betas=(0.9, 0.99))
    hist = {'g': [], 's': [], 'joint': []}
    acc = metrics.Evaluator(2)
    kw = dict(perception_weight=perception_weight, ssim_weight=ssim_weight)

    def batches(epoch):
        return tiles.Prefetcher(_loader(ds, batch_size, True, seed * 1000 + epoch), dev)

    for ep in range(epochs_g):                                                   # Demo_USSS.py:126-189
        tot = torch.zeros((), device=dev)
        for x, y, item, r in batches(ep):
            out = steps.usss_g_pretrain_step(netG, crit, optG, x, y, **kw)
            tot += out['loss'].detach() * x.shape[0] / len(ds)
        hist['g'].append(float(tot))
        if log:
            log('G pre-train epoch %d loss %.4f' % (ep + 1, hist['g'][-1]))
    for ep in range(epochs_s):                                                   # Demo_USSS.py:194-286
        tot = torch.zeros((), device=dev)
        for x, y, item, r in batches(1000 + ep):
            out = steps.usss_s_pretrain_step(netS, netG, crit, optS, x, y, l1_weight=l1_weight, **kw)
            tot += out['net_loss'].detach() * x.shape[0] / len(ds)
        hist['s'].append(float(tot))
        if log:
            log('S pre-train epoch %d loss %.4f' % (ep + 1, hist['s'][-1]))
    for ep in range(epochs_joint):                                               # Demo_USSS.py:291-400
        tot = torch.zeros((), device=dev)
        acc.reset()
        for x, y, item, r in batches(2000 + ep):
            out = steps.usss_joint_step(netS, netG, crit, optS, optG, x, y, l1_weight=l1_weight, **kw)
            tot += out['net_loss'].detach() * x.shape[0] / len(ds)
            if ref is not None:
                acc.add_batch_map(r, metrics.threshold_map(out['cmap'].detach(), prob_thresh), gt_map, pre_map)
        hist['joint'].append(float(tot))
        if log:
            log('joint epoch %d loss %.4f' % (ep + 1, hist['joint'][-1]))

    res = infer_scene(netS, ds, dev, batch_size=batch_size, prob_thresh=prob_thresh, gt_map=gt_map, pre_map=pre_map)
    if out_density:
        tiles.write_tiff(out_density, res['density'])
    if out_color:
        tiles.write_tiff(out_color, res['color'])
    res.update(netS=netS, netG=netG, history=hist, train_evaluator=acc)
    return res


@torch.no_grad()
def infer_scene(netS, ds, device, batch_size=10, prob_thresh=0.5, gt_map=(1, 2), pre_map=(0, 1), eval_mode=True):
    """Inference over every tile of a PairTileDataset with centre write-back
    (Demo_USSS.py:404-470, Demo_RSSS.py:451-491): density map (float32), TP/FP/FN colour codes
    and the evaluator over the owned centres.  ``eval_mode=False`` keeps train() statistics
    like Demo_WSSS.py:389-391."""
    dev = torch.device(device)
    was_training = netS.training
    netS.train(not eval_mode)
    grid = ds.grid
    density = np.zeros((1, grid.ysize, grid.xsize), np.float32)
    color = np.zeros((1, grid.ysize, grid.xsize), np.float32)
    acc = metrics.Evaluator(2)
    for x, y, item, r in tiles.Prefetcher(DataLoader(ds, batch_size=batch_size, shuffle=False), dev):
        cmap, mask = steps.infer_density(netS, x, y, prob_thresh)
        maskf = mask.to(cmap.dtype)
        valid = torch.zeros_like(mask)
        for i, it in enumerate(item.tolist()):
            r0, r1, c0, c1 = grid.eff_range(it)
            valid[i, :, r0:r1, c0:c1] = True
        if ds.ref is not None:
            acc.add_batch_map(r, maskf, gt_map, pre_map, valid=valid)
            codes = metrics.changemap_codes(maskf, r, True, ref_map=gt_map, dt_map=pre_map)
        else:
            codes = metrics.changemap_codes(maskf, maskf, False, dt_map=pre_map)
        cm_h, co_h = cmap.cpu().numpy(), codes.cpu().numpy()
        for i, it in enumerate(item.tolist()):
            grid.write_center(density, cm_h[i], it)
            grid.write_center(color, co_h[i], it)
    netS.train(was_training)
    return dict(density=density, color=color, evaluator=acc)
