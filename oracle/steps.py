"""Oracle (test infrastructure): the train-step bodies of Demo_USSS / Demo_RSSS /
Demo_WSSS restated over the functional nets (``oracle.nets``) with the *literal*
call order of the reference (retain_graph double-backward, zero_grad placement),
stepping stock ``torch.optim`` optimizers.

State container ``Nets``: dict of state_dicts {'G','S','D','VGG'} whose parameter
tensors require grad, plus the optimizers that own them.
"""
import torch

from . import nets, losses


class Nets:
    def __init__(self, sdG=None, sdS=None, sdD=None, sdVGG=None, bilinear=True):
        self.G = nets.clone_state(sdG) if sdG is not None else None
        self.S = nets.clone_state(sdS) if sdS is not None else None
        self.D = nets.clone_state(sdD) if sdD is not None else None
        self.VGG = nets.clone_state(sdVGG, requires_grad=False) if sdVGG is not None else None
        self.bilinear = bilinear
        self.opt = {}
        self.capture = None     # set to {} to record the gradients each optimizer is about to step on

    def step(self, which):
        """optimizer.step() of net ``which``; with ``capture`` set, first snapshots name -> gradient
        exactly as the optimizer sees it (the parity tests compare the HIP path's pre-step gradients)."""
        if self.capture is not None:
            sd = getattr(self, which)
            self.capture[which] = {k: (sd[k].grad.detach().clone() if sd[k].grad is not None else None)
                                   for k in nets.param_keys(sd)}
        self.opt[which].step()

    def params(self, which):
        sd = getattr(self, which)
        return [sd[k] for k in nets.param_keys(sd)]

    def make_optimizers(self, kind):
        """Optimizer line-up of each demo:
        USSS  Demo_USSS.py:121-122  Adam(2e-4,(0.9,0.99)) for S and G
        RSSS  Demo_RSSS.py:151-158  G Adam(5e-5,(0.9,0.99)); S, D RMSprop(5e-5)
        WSSS  Demo_WSSS.py:116-122  G Adam(5e-4,(0.9,0.99)); S RMSprop(1e-3); D RMSprop(1e-5)
        """
        if kind == 'usss':
            self.opt['G'] = torch.optim.Adam(self.params('G'), lr=2e-4, betas=(0.9, 0.99))
            self.opt['S'] = torch.optim.Adam(self.params('S'), lr=2e-4, betas=(0.9, 0.99))
        elif kind == 'rsss':
            self.opt['G'] = torch.optim.Adam(self.params('G'), lr=5e-5, betas=(0.9, 0.99))
            self.opt['S'] = torch.optim.RMSprop(self.params('S'), lr=5e-5)
            self.opt['D'] = torch.optim.RMSprop(self.params('D'), lr=5e-5)
        elif kind == 'wsss':
            self.opt['G'] = torch.optim.Adam(self.params('G'), lr=5e-4, betas=(0.9, 0.99))
            self.opt['S'] = torch.optim.RMSprop(self.params('S'), lr=1e-3)
            self.opt['D'] = torch.optim.RMSprop(self.params('D'), lr=1e-5)
        else:
            raise ValueError(kind)
        return self


def adjust_learning_rate(optimizer, epoch, lr_start=1e-4, lr_max=1e-3, lr_min=1e-6,
                         lr_warm_up_epoch=20, lr_sustain_epochs=0, lr_exp_decay=0.8):
    """CommonFunc.py:23-37 -- linear warm-up, sustain, exponential decay."""
    if epoch < lr_warm_up_epoch:
        lr = (lr_max - lr_start) / lr_warm_up_epoch * epoch + lr_start
    elif epoch < lr_warm_up_epoch + lr_sustain_epochs:
        lr = lr_max
    else:
        lr = (lr_max - lr_min) * lr_exp_decay ** (epoch - lr_warm_up_epoch - lr_sustain_epochs) + lr_min
    for g in optimizer.param_groups:
        g['lr'] = lr
    return lr


# ------------------------------------------------------------------- USSS
def usss_g_pretrain_step(n, x, y, perception_weight=0.4, ssim_weight=0):
    """Demo_USSS.py:142-159."""
    n.opt['G'].zero_grad()
    y_fake = nets.generator(n.G, x, train=True)
    cmap = torch.zeros((x.shape[0], 1, x.shape[2], x.shape[3]), dtype=x.dtype, device=x.device)
    gen, l1, perc, ssim = losses.cnet_loss(n.VGG, y, y_fake, cmap)
    loss = gen + perception_weight * perc + ssim_weight * ssim
    loss.backward()
    n.step('G')
    return dict(loss=loss, gen=gen, perc=perc, ssim=ssim, y_fake=y_fake)


def usss_s_pretrain_step(n, x, y, perception_weight=0.4, l1_weight=0.65, ssim_weight=0):
    """Demo_USSS.py:219-228 (G forward in train mode, G never stepped)."""
    y_fake = nets.generator(n.G, x, train=True)
    cmap = nets.segmentor(n.S, x, y, train=True, bilinear=n.bilinear)
    gen, l1, perc, ssim = losses.cnet_loss(n.VGG, y, y_fake, cmap)
    net_loss = gen + l1_weight * l1 + perception_weight * perc + ssim_weight * ssim
    n.opt['S'].zero_grad()
    net_loss.backward()
    n.step('S')
    return dict(net_loss=net_loss, gen=gen, l1=l1, perc=perc, ssim=ssim, cmap=cmap)


def usss_joint_step(n, x, y, perception_weight=0.4, l1_weight=0.65, ssim_weight=0):
    """Demo_USSS.py:310-341 -- two backward passes over one graph: G ends up with
    grad(Loss)+grad(NetLoss), S only with grad(NetLoss)."""
    n.opt['G'].zero_grad()
    y_fake = nets.generator(n.G, x, train=True)
    cmap = nets.segmentor(n.S, x, y, train=True, bilinear=n.bilinear)
    gen, l1, perc, ssim = losses.cnet_loss(n.VGG, y, y_fake, cmap)
    loss = gen + perception_weight * perc + ssim_weight * ssim
    loss.backward(retain_graph=True)
    net_loss = gen + l1_weight * l1 + perception_weight * perc + ssim_weight * ssim
    n.opt['S'].zero_grad()
    net_loss.backward()
    n.step('G')
    n.step('S')
    return dict(loss=loss, net_loss=net_loss, gen=gen, l1=l1, perc=perc, ssim=ssim, cmap=cmap)


# ------------------------------------------------------------------- RSSS
def rsss_g_pretrain_step(n, x, y, region, perception_weight=0.1, ssim_weight=0, per_band=True):
    """Demo_RSSS.py:190-208 (region plays the role of cmap)."""
    n.opt['G'].zero_grad()
    y_fake = nets.generator(n.G, x, train=True)
    gen, ssim, perc = losses.cgenerator_loss(n.VGG, y, y_fake, region, 1, per_band)
    g_loss = gen + perception_weight * perc + ssim_weight * ssim
    g_loss.backward()
    n.step('G')
    return dict(g_loss=g_loss, gen=gen, perc=perc, ssim=ssim)


def rsss_adversarial_step(n, x, y, region, perception_weight=0.1, ssim_weight=0, per_band=True,
                          l1_weight=0.02, g_weight=0.5, d_weight=1, r_weight=2,
                          discriminator_continuous=True):
    """Demo_RSSS.py:285-332, literal order.  netG in eval mode (Demo_RSSS.py:240)."""
    C = x.shape[1]
    cmap = nets.segmentor(n.S, x, y, train=True, bilinear=n.bilinear)
    cmask = cmap if discriminator_continuous else (torch.sign(cmap - 0.5) + 1) / 2
    keep = 1 - cmask.repeat((1, C, 1, 1))
    x_mask, y_mask = x * keep, y * keep
    c_out = nets.discriminator(n.D, x_mask, y_mask, train=True)
    y_unc = y * (1 - region) + x * region
    nc_out = nets.discriminator(n.D, x * keep, y_unc * keep, train=True)
    n.opt['D'].zero_grad()
    d_loss = 1 + nc_out.mean() - c_out.mean()
    d_loss.backward(retain_graph=True)
    n.step('D')

    c_out = nets.discriminator(n.D, x_mask, y_mask, train=True)
    y_fake = nets.generator(n.G, x, train=False)
    gen, ssim, perc = losses.cgenerator_loss(n.VGG, y, y_fake, cmap, 1, per_band)
    g_loss = gen + perception_weight * perc + ssim_weight * ssim
    l1_loss = losses.region_loss(cmap, region, 'l1')
    s_d_loss = c_out.mean()
    r_loss = losses.region_loss(cmap, 1 - region, 'mse')
    s_loss = d_weight * s_d_loss + l1_weight * l1_loss + g_weight * g_loss + r_weight * r_loss
    n.opt['S'].zero_grad()
    s_loss.backward()
    n.step('S')
    return dict(d_loss=d_loss, s_loss=s_loss, s_d_loss=s_d_loss, g_loss=g_loss, l1_loss=l1_loss,
                r_loss=r_loss, gen=gen, ssim=ssim, perc=perc, cmap=cmap)


# ------------------------------------------------------------------- WSSS
def wsss_adversarial_step(n, x, y, x_nc, y_nc, perception_weight=0.5, ssim_weight=0,
                          g_weight=0.2, l1_weight=1.6, d_weight=1, nc_weight=1.5,
                          discriminator_continuous=True):
    """Demo_WSSS.py:249-323, literal order.  netG in eval mode (Demo_WSSS.py:206).
    The unchanged pair is masked with the CHANGED pair's map (:278-279)."""
    C = x.shape[1]
    cmap = nets.segmentor(n.S, x, y, train=True, bilinear=n.bilinear)
    cmask = cmap if discriminator_continuous else (torch.sign(cmap - 0.5) + 1) / 2
    keep = 1 - cmask.repeat((1, C, 1, 1))
    x_mask, y_mask = x * keep, y * keep
    c_out = nets.discriminator(n.D, x_mask, y_mask, train=True)
    ncmap = nets.segmentor(n.S, x_nc, y_nc, train=True, bilinear=n.bilinear)
    nc_out = nets.discriminator(n.D, x_nc * keep, y_nc * keep, train=True)
    n.opt['D'].zero_grad()
    d_loss = 1 + nc_out.mean() - c_out.mean()
    d_loss.backward(retain_graph=True)
    n.step('D')

    nc_loss = torch.mean(torch.pow(ncmap, 2))
    c_out = nets.discriminator(n.D, x_mask, y_mask, train=True)
    y_fake = nets.generator(n.G, x, train=False)
    gen, ssim, perc = losses.cgenerator_loss(n.VGG, y, y_fake, cmap, 1, False)
    g_loss = gen + perception_weight * perc + ssim_weight * ssim
    l1_loss = torch.mean(abs(cmap))
    s_d_loss = c_out.mean()
    s_loss = d_weight * s_d_loss + l1_weight * l1_loss + g_weight * g_loss + nc_weight * nc_loss
    n.opt['S'].zero_grad()
    s_loss.backward()
    n.step('S')
    return dict(d_loss=d_loss, s_loss=s_loss, s_d_loss=s_d_loss, g_loss=g_loss, l1_loss=l1_loss,
                nc_loss=nc_loss, gen=gen, ssim=ssim, perc=perc, cmap=cmap, ncmap=ncmap)
