#!/usr/bin/env python3
"""where does the Generator-step gradient's distance to the fp64 truth on the Winograd plan come from --
ReLU / max-pool DECISIONS the F(4x4) forward rounding turns the other way, or the F(4x4) arithmetic of the backward pass?

Demo_USSS Generator pre-training step (Demo_USSS.py:142-159), 4 bands 256 x 256, 2 tiles, as tests/test_gpu_fullsize_bwd.py runs it.
The conv plan is a run-time switch of the library (fcd_conv_wino_set) and every backward kernel takes its ReLU mask / pool code from
what the FORWARD pass saved, so the two halves can be mixed:
    forward plan  x  backward plan   in {direct, winograd}^2
 - (direct forward, winograd backward): the decisions of the accurate forward (stock-fp32 level), F(4x4) arithmetic backward
 - (winograd forward, direct backward): the F(4x4) forward's decisions, direct-kernel arithmetic backward
Prints the relative L2 distance of G's whole gradient to the fp64 gradient for the four combinations, next to stock fp32."""
import json
import os
import sys
import warnings

import torch

# every layer saves its decisions as fp32 masks / pool codes and is its own autograd node: the backward kernel of a layer is then
# free to follow the plan in force at backward time (the bit-mask and chain nodes are F(4x4)-only; same decisions, bit for bit)
os.environ['FCD_WINO_CHAIN'] = '0'
os.environ['FCD_WINO_RELU_BITS'] = '0'

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

from seeded import seeded_state, seeded_tiles          # noqa: E402
from oracle import nets as onets, steps as osteps      # noqa: E402
import test_gpu_fullsize_bwd as T                      # noqa: E402


from decisions import inject_direct_vgg_decisions          # noqa: E402  (tests/decisions.py)


def main():
    import fcd_gan_pytorch_amd as p
    from fcd_gan_pytorch_amd import _lib, _ops as ops
    C, N, H = 4, 2, 256
    sdG = seeded_state(onets.generator_spec(C), 41)
    sdV = seeded_state(onets.vgg_spec(), 4242)
    x, y, _ = seeded_tiles(43, N, C, H, H)
    n = osteps.Nets(sdG, None, None, sdV)
    n.opt['G'] = torch.optim.Adam(n.params('G'), lr=2e-4, betas=(0.9, 0.99))
    n.capture = {}
    osteps.usss_g_pretrain_step(n, x, y)

    def run64(m):
        m.opt['G'] = torch.optim.Adam(m.params('G'), lr=2e-4, betas=(0.9, 0.99))
        osteps.usss_g_pretrain_step(m, T._t64(x), T._t64(y))
    g64 = T._oracle_fp64((sdG, None, None, sdV), None, run64)['G']
    names = [k for k in g64 if g64[k] is not None and not T.is_pre_bn_bias(k)]
    flat64 = torch.cat([g64[k].reshape(-1) for k in names])
    o32 = torch.cat([n.capture['G'][k].double().reshape(-1) for k in names])
    out = {'oracle32': ((o32 - flat64).norm() / flat64.norm()).item()}
    lib = _lib.lib
    prev = lib.fcd_conv_wino_set(-1)
    try:
        for fwd in (0, 4):
            for bwd in (0, 4):
                lib.fcd_conv_wino_set(fwd)
                netG = p.Module.Generator(C)
                netG.load_state_dict(sdG)
                with warnings.catch_warnings():
                    warnings.simplefilter('ignore')
                    crit = p.Loss.CNetLoss(channel=C, perception_layer=1, perception_perBand=True, allow_seeded=True)
                crit.loss_perception.net.load_state_dict(sdV)
                crit.to('cuda')
                netG.to('cuda').train()
                xg, yg = x.cuda(), y.cuda()
                y_fake = netG(xg)
                cmap = torch.zeros((N, 1, H, H), device='cuda')
                gen, l1, perc, ssim = crit(yg, y_fake, cmap)
                loss = gen + 0.4 * perc
                lib.fcd_conv_wino_set(bwd)          # the backward kernels follow the plan in force NOW; masks / codes are the forward's
                loss.backward()
                torch.cuda.synchronize()
                got = dict(netG.named_parameters())
                flat = torch.cat([got[k].grad.detach().cpu().double().reshape(-1) for k in names])
                key = 'fwd_%s__bwd_%s' % ('winograd' if fwd else 'direct', 'winograd' if bwd else 'direct')
                out[key] = ((flat - flat64).norm() / flat64.norm()).item()
        # Winograd plan everywhere, but every VGG layer's ReLU mask / max-pool code taken from the direct kernels' forward
        lib.fcd_conv_wino_set(4)
        with inject_direct_vgg_decisions():
            netG = p.Module.Generator(C)
            netG.load_state_dict(sdG)
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                crit = p.Loss.CNetLoss(channel=C, perception_layer=1, perception_perBand=True, allow_seeded=True)
            crit.loss_perception.net.load_state_dict(sdV)
            crit.to('cuda')
            netG.to('cuda').train()
            y_fake = netG(x.cuda())
            gen, l1, perc, ssim = crit(y.cuda(), y_fake, torch.zeros((N, 1, H, H), device='cuda'))
            (gen + 0.4 * perc).backward()
            torch.cuda.synchronize()
            got = dict(netG.named_parameters())
            flat = torch.cat([got[k].grad.detach().cpu().double().reshape(-1) for k in names])
            out['winograd_values__direct_vgg_decisions'] = ((flat - flat64).norm() / flat64.norm()).item()
    finally:
        lib.fcd_conv_wino_set(prev)
    out['ratios_to_oracle32'] = {k: v / out['oracle32'] for k, v in out.items() if k != 'oracle32'}
    print(json.dumps(out, indent=1))
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'parity_probe_g.json'), 'w') as f:
        json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
