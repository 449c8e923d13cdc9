"""BACKWARD parity at BASELINE.json's tile sizes: one whole train iteration of each demo on the HIP path
against the CPU oracle step (oracle/steps.py, literal reference order), comparing

 (a) the gradient of every stepped network exactly as its optimizer sees it (pre-step hook on the fcd
     optimizer / ``Nets.capture`` on the oracle): relative L2 of the whole flat gradient and of every
     parameter tensor that carries more than 1e-4 of the largest tensor norm,
 (b) the weights after the optimizer step, per tensor: the applied update against the oracle's, and the
     HIP update kernel against torch.optim's rule applied to the SAME gradient (isolates the optimizer),
 (c) BatchNorm running statistics and num_batches_tracked after the iteration,

on both conv plans (``conv_path``: direct MFMA kernels only / Winograd F(4x4,3x3) on the wide layers).
Shapes: configs[2] Demo_RSSS 13 bands 256x256 (Demo_RSSS.py:285-332), configs[1] Demo_USSS generator step
4 bands 256x256 (Demo_USSS.py:142-159), configs[4] Demo_WSSS 3 bands 512x512 (Demo_WSSS.py:249-323).

Bounds come from an fp64 TRUTH (round 3): next to the fp32 oracle step the SAME oracle step runs once
in double precision (``torch.set_default_dtype(float64)``, weights / inputs widened exactly), and every gradient is
judged by its distance to that truth RELATIVE to the fp32 oracle's own distance to it:

    ||g_HIP - g64||  <=  K * ||g_oracle32 - g64||  + floor * ||g64||        (flat, and per parameter tensor)

i.e. the HIP path may be at most K times as far from the exact gradient as stock fp32 PyTorch on the CPU is; no bound is
derived from a perturbation experiment any more.  K, floors and the measured ratios: ``K_TRUTH`` below and
profiles/r03_parity_fullsize.md (both errors side by side).

The Discriminator step is different, and the fp64 machinery is what shows how (tools/parity_probe_d.py, table in
profiles/r03_parity_fullsize.md).  d_loss = 1 + mean D(unchanged) - mean D(changed), D(a, b) = classifier(net(a) - net(b)):
differences of nearly equal terms at two levels.
 (1) Its inputs are masked with the Segmentor's density map, and in fp64 a 1e-5 change of that map moves D's gradient by
     2e-3 ... 1e-2 (condition number 200 ... 1000, measured per run below).  The HIP Winograd plan's map is 2e-5 from the
     exact one (inside the 1e-4 the spec allows, asserted below), stock fp32's 1e-6.  So D is judged against the fp64
     D-step EVALUATED ON THE MAP THE PATH ACTUALLY PRODUCED, G64_D(cmap_HIP) -- its own arithmetic, not the Segmentor's.
 (2) Even on identical inputs ANY fp32 evaluation of this gradient lands 3e-6 ... 1e-2 from the fp64 value depending on
     rounding-level details of the input: on four maps that differ by <= 2e-5 the CPU oracle (oneDNN) is 6.0e-4, 2.6e-4,
     3.6e-6, 1.0e-3 off and the HIP kernels 3.0e-6, 3.9e-3, 1.4e-3, 1.2e-2 -- neither is "the accurate one", the ratio
     between them swings over five decades, and a rule  err_HIP <= K err_oracle32  would pass or fail by chance.  D's
     gradient is therefore held to an ABSOLUTE distance from G64_D(own map): 2e-2 flat, 3e-2 per tensor (HIP_D_LIMITS), and
     both fp32 errors plus the end-to-end figures are written to the report.
S and G (no difference of twin terms) are judged end to end with the ratio rule.
Conv biases that feed a BatchNorm are excluded from (a)/(b): their true gradient is exactly zero, and what any
implementation computes there is rounding noise (checked to be small against the weight gradients instead).  D's
BatchNorm statistics include a forward pass AFTER its sign-like RMSprop update, so they inherit the update's
sensitivity (bound 5e-3; measured 2.7e-4 ... 2.3e-3); S's and G's are updated before any step (bound 1e-4).
"""
import json
import os
import warnings

import numpy as np
import pytest
import torch

from seeded import seeded_state, seeded_tiles
from oracle import nets as onets, steps as osteps
from test_gpu_modules import is_pre_bn_bias

pytestmark = pytest.mark.gpu
DEV = 'cuda'

# Bounds (keys of the report each net gets).  See DESIGN.md section 2 for the measured values they are set from.
LIMITS = {
    'flat_over_rule': 1.0,                    # ||g_HIP - g64|| / (K ||g_o32 - g64|| + floor ||g64||), whole flat gradient
    'worst_tensor_over_rule': 1.0,            # ... worst parameter tensor (non-scalar, above 1e-4 of the largest norm)
    'pre_bn_bias_grad_max_over_wmax': 1e-3,   # analytically-zero gradients stay at rounding level
    'update_kernel_vs_torch_rule': 1e-4,      # HIP update kernel vs torch.optim's rule on the same gradient (beyond 1 ulp)
    'worst_update_rel_l2': 2e-2,              # applied update vs the oracle's over sign-settled elements
    'max_weight_diff_over_step': 2.05,        # both moved by at most one step size
    'bn_running_rel_err': 1e-4,
}


# ||g_HIP - g64|| <= K * ||g_o32 - g64|| + floor * ||g64||.  Measured on MI355X (profiles/r03_parity_fullsize.md):
# see the table there; K = 2 on the direct plan, and on the Winograd plan K = 2 for everything but the tensors listed
# with their measured ratios in that file (F(4x4) transforms carry ~1e-5 rounding per layer vs ~1e-6 direct).
K_TRUTH = {'direct': dict(k_flat=2.0, k_tensor=3.0, floor_flat=2e-4, floor_tensor=5e-4),
           'winograd': dict(k_flat=6.0, k_tensor=10.0, floor_flat=2e-4, floor_tensor=5e-4),
           # per net where the measured ratio leaves room (~1.5x the measurement, so that a regression in the blocked-M /
           # transposed-B GEMM / BatchNorm-statistics kernels cannot hide in the Generator step's slack): Segmentor on the Winograd
           # plan measured 1.9 - 2.0 flat, 3.1 worst tensor
           ('winograd', 'S'): dict(k_flat=3.0, k_tensor=5.0, floor_flat=2e-4, floor_tensor=5e-4),
           # [r5] The Generator step's gradient runs back through the 13 frozen VGG layers of the perception term.  Its distance to the
           # fp64 gradient is set by which ReLU / max-pool DECISIONS the forward pass turns the other way, not by the arithmetic:
           # measured (tools/parity_probe_g.py, test_usss_generator_gradient_with_direct_vgg_decisions below) -- backward plan irrelevant
           # (direct forward + F(4x4) backward 1.043x stock fp32, direct + direct 1.042x), F(4x4) forward 5.42x, and F(4x4) forward AND
           # backward with only the VGG masks / pool codes taken from the direct forward 1.12x.  A flipped unit contributes its whole dy to
           # the error, so the relative gradient error goes like sqrt(fraction of flipped units) ~ sqrt(forward rounding): F(4x4)'s ~30x
           # coarser forward rounding (1.4e-5 vs ~5e-7 of a layer output) gives ~5.4x.  The ARITHMETIC is held to 2x by the injected-
           # decisions test; the end-to-end figure, which measures decision sensitivity, to 8 / 13 (final builds of rounds 5 and 6 measure
           # 5.49 flat / 9.33 worst tensor, profiles/r06_parity_fullsize.md; 5.42 / 9.03 before the round-5 chain kernels).
           ('winograd', 'G'): dict(k_flat=8.0, k_tensor=13.0, floor_flat=2e-4, floor_tensor=5e-4)}
# (Winograd plan, measured: the generator's gradient through the 13 F(4x4) VGG layers of the perception term ends 4.4x
#  (flat) / 8.1x (worst tensor) as far from the fp64 truth as stock fp32 -- 1.5e-3 / 2.5e-3 absolute; the Segmentor 1.9x /
#  3.1x.  Direct plan: 1.0 - 1.5x flat, <= 2.5x per tensor.)
# D against G64_D(own map).  test_discriminator_step_gradient_error_distribution below measures, on 16 maps, how far ANY fp32
# evaluation lands from the fp64 D gradient: either ~3e-6 (no activation decision differs) or a discrete jump of 2e-4 ... 1.05e-2
# (one LeakyReLU / BatchNorm-ed unit of the 4 x 1024-unit classifier or of net falls on the other side), for the CPU oracle and the
# HIP kernels alike (both max 1.05e-2).  A single draw is therefore held to 1.5 x the largest jump the CPU oracle itself shows;
# the comparison of the two implementations is the distribution test's job.
D_LARGEST_JUMP = 1.05e-2
HIP_D_LIMITS = dict(flat=1.5 * D_LARGEST_JUMP, tensor=2.5 * D_LARGEST_JUMP)


# Where the fp64 TRUTH runs: the oracle's stock torch ops in double precision on the DEVICE (ATen / rocBLAS fp64 kernels -- as foreign
# to the product's HIP kernels as oneDNN is).  Measured on the MI355X box (tools/debug, round 6): the fp64 Demo_RSSS oracle step at
# 13 x 256 x 256, N = 2 takes 2.2 s there against 74 s on the 16 host cores, and the two sets of fp64 gradients agree to 3e-14 (S) /
# 4e-14 (D) relative L2, the density map to 8e-14 -- the truth does not care where it is evaluated, the suite's wall clock does
# (VERDICT r5 weak 11).  The fp32 ORACLE -- the yardstick "how far is stock fp32 PyTorch on the reference's CPU path from the truth" --
# stays on the host.
TRUTH_DEV = 'cuda'


def _t64(t):
    return t.double().to(TRUTH_DEV)


def _dbl(sd):
    return None if sd is None else {k: (v.double() if v.is_floating_point() else v.clone()).to(TRUTH_DEV) for k, v in sd.items()}


def _oracle_fp64(sds, kind, run):
    """The oracle step in double precision: returns {net: {name: fp64 gradient}} as its optimizers see them."""
    from oracle import losses as olosses
    prev, prev_bb = torch.get_default_dtype(), olosses.BATCH_BANDS
    torch.set_default_dtype(torch.float64)
    olosses.BATCH_BANDS = True        # same function, the per-band VGG passes as one batch (oracle/losses.py): ~3x less wall time in fp64
    try:
        n = osteps.Nets(*[_dbl(sd) for sd in sds])
        if kind:
            n.make_optimizers(kind)
        n.capture = {}
        out = run(n)
        for which in n.capture:
            for k, g in n.capture[which].items():
                assert g is None or g.dtype == torch.float64, (which, k, g.dtype)
            n.capture[which] = {k: (None if g is None else g.cpu()) for k, g in n.capture[which].items()}
        if isinstance(out, dict) and 'cmap' in out:
            n.capture['cmap64'] = out['cmap'].detach().cpu()
        return n.capture
    finally:
        torch.set_default_dtype(prev)
        olosses.BATCH_BANDS = prev_bb


def _d_step_fp64(sdD, c_pair, nc_pair):
    """G64_D: gradients of d_loss = 1 + mean D(nc) - mean D(c) (Demo_RSSS.py:292-305, Demo_WSSS.py:268-285) in double
    precision for given (already masked) input pairs."""
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        oD = onets.clone_state(_dbl(sdD))
        c_out = onets.discriminator(oD, _t64(c_pair[0]), _t64(c_pair[1]), train=True)
        nc_out = onets.discriminator(oD, _t64(nc_pair[0]), _t64(nc_pair[1]), train=True)
        (1 + nc_out.mean() - c_out.mean()).backward()
        return {k: oD[k].grad.detach().cpu() for k in onets.param_keys(oD)}
    finally:
        torch.set_default_dtype(prev)


def _flat64(g, keys):
    return torch.cat([g[k].reshape(-1).double() for k in keys])


def _d_own_truths(d_truth, g64, cm_hip, cm_o32, cache):
    """(G64_D(cmap_HIP), G64_D(cmap_oracle32), measured amplification of a map deviation into D's gradient)."""
    if 'o32' not in cache:
        cache['o32'] = d_truth(cm_o32)
    th = d_truth(cm_hip)
    keys = [k for k in g64['D'] if not is_pre_bn_bias(k)]
    dg = (_flat64(th, keys) - _flat64(g64['D'], keys)).norm().item() / _flat64(g64['D'], keys).norm().item()
    dm = (cm_hip.double() - g64['cmap64']).abs().max().item()
    print('\n[D conditioning] map deviation (HIP vs fp64) %.2e moves G64_D by %.2e relative: amplification %.0f' % (dm, dg, dg / max(dm, 1e-30)))
    return th, cache['o32'], dict(map_dev=dm, grad_rel_change=dg, amplification=dg / max(dm, 1e-30))


def _truth_limits(o32, g64, which, d_bn=False):
    """Limits that still refer to the fp32 oracle's post-step STATE (applied update, BatchNorm statistics): scaled by
    the fp32 oracle's own flat distance to the fp64 truth."""
    keys = [k for k in g64[which] if not is_pre_bn_bias(k)]
    flat = rl2(torch.cat([o32[which][k].reshape(-1) for k in keys]), torch.cat([g64[which][k].reshape(-1) for k in keys]))
    lim = {'worst_update_rel_l2': max(2e-2, 6 * flat)}
    if d_bn:
        lim['bn_running_rel_err'] = 5e-3     # D's statistics include a forward pass AFTER its sign-like update; measured 2.7e-4 ... 2.3e-3
    print('\n[fp32 oracle vs fp64 truth, %s] flat %.2e' % (which, flat))
    return lim


_ORACLE = {}             # config -> oracle result (shared by the two conv_path runs)
_REPORT = {}


def pkg():
    import fcd_gan_pytorch_amd as p
    return p


def rl2(a, b):
    a, b = a.detach().cpu().double().reshape(-1), b.detach().cpu().double().reshape(-1)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _hook(store, key):
    def f(opt):
        store[key] = opt.flat_g.detach().clone()
        store[key + '/p_before'] = opt.flat_p.detach().clone()
    return f


def _named_slices(net):
    out, off = [], 0
    for k, p in net.named_parameters():
        out.append((k, off, p.numel(), tuple(p.shape)))
        off += p.numel()
    return out


def _torch_rule(kind, p, g, lr):
    """First optimizer step of torch.optim.RMSprop(alpha .99, eps 1e-8) / Adam(betas (.9,.99), eps 1e-8) on CPU."""
    p, g = p.cpu().clone(), g.cpu()
    prm = torch.nn.Parameter(p)
    prm.grad = g.clone()
    opt = torch.optim.RMSprop([prm], lr=lr) if kind == 'rmsprop' else torch.optim.Adam([prm], lr=lr, betas=(0.9, 0.99))
    opt.step()
    return prm.detach()


def check_net(tag, which, net, opt_kind, lr, store, oracle_grads, oracle_sd, truth, plan, limits=None, truth_own=None):
    """(a) + (b) + (c) for one stepped network."""
    g_got, p_before = store[which].cpu(), store[which + '/p_before'].cpu()
    p_after = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu()
    slices = _named_slices(net)
    keep = [s for s in slices if not is_pre_bn_bias(s[0])]
    cat = lambda t: torch.cat([t[o:o + n] for _, o, n, _ in keep])
    g_ref_full = torch.cat([oracle_grads[k].reshape(-1) for k, _, _, _ in slices])
    rep = {}
    # (a) gradients against the fp64 truth, next to the fp32 oracle's own distance to it
    # truth_own = (G64(forward state of the HIP path), G64(forward state of the fp32 oracle)): see the module docstring
    kt = K_TRUTH.get((plan, which), K_TRUTH[plan])
    g64_full = torch.cat([truth[k].reshape(-1) for k, _, _, _ in slices])
    n64 = cat(g64_full).double().norm().item()
    if truth_own is not None:
        rep['e2e_flat_rel_l2_vs_fp64'] = (cat(g_got).double() - cat(g64_full)).norm().item() / n64
        rep['e2e_flat_rel_l2_oracle32_vs_fp64'] = (cat(g_ref_full).double() - cat(g64_full)).norm().item() / n64
        truth_h, truth_o = truth_own[0], truth_own[1]
        rep['amplification_of_forward_map_deviation'] = truth_own[2] if len(truth_own) > 2 else None
    else:
        truth_h = truth_o = truth
    gh_full = torch.cat([truth_h[k].reshape(-1) for k, _, _, _ in slices])
    go_full = torch.cat([truth_o[k].reshape(-1) for k, _, _, _ in slices])
    eH, eO = (cat(g_got).double() - cat(gh_full)).norm().item(), (cat(g_ref_full).double() - cat(go_full)).norm().item()
    rep['flat_rel_l2_vs_fp64'], rep['flat_rel_l2_oracle32_vs_fp64'] = eH / n64, eO / n64
    rep['flat_rel_l2'] = rl2(cat(g_got), cat(g_ref_full))                       # vs the fp32 oracle (informational)
    rep['flat_over_rule'] = eH / (kt['k_flat'] * eO + kt['floor_flat'] * n64) if truth_own is None else eH / (HIP_D_LIMITS['flat'] * n64)
    norms = {k: truth[k].norm().item() for k, _, _, _ in keep}
    nmax = max(norms.values())
    worst, worst_k, worst_pair = 0.0, None, (0.0, 0.0)
    worst_ratio, worst_ratio_k = 0.0, None
    for k, o, n, _ in keep:
        if norms[k] > 1e-4 * nmax and n > 1:          # (scalars -- PReLU slopes -- only enter the flat norm: a sum of
            eh = (g_got[o:o + n].double() - truth_h[k].reshape(-1)).norm().item()    #  64 x H x W signed terms is one ill-conditioned number)
            eo = (oracle_grads[k].reshape(-1).double() - truth_o[k].reshape(-1)).norm().item()
            v = eh / (kt['k_tensor'] * eo + kt['floor_tensor'] * norms[k]) if truth_own is None else eh / (HIP_D_LIMITS['tensor'] * norms[k])
            if v > worst:
                worst, worst_k, worst_pair = v, k, (eh / norms[k], eo / norms[k])
            if eh / max(eo, 1e-30) > worst_ratio and eh > kt['floor_tensor'] * norms[k]:
                worst_ratio, worst_ratio_k = eh / max(eo, 1e-30), k
    rep['worst_tensor_over_rule'], rep['worst_tensor'] = worst, worst_k
    rep['worst_tensor_rel_l2_vs_fp64'], rep['worst_tensor_oracle32_vs_fp64'] = worst_pair
    rep['worst_error_ratio_above_floor'], rep['worst_error_ratio_tensor'] = worst_ratio, worst_ratio_k
    norms = {k: oracle_grads[k].double().norm().item() for k, _, _, _ in keep}
    nmax = max(norms.values())
    wmax = max(oracle_grads[k].abs().max().item() for k, _, _, _ in keep if k.endswith('weight'))
    zero_bias = max((g_got[o:o + n].abs().max().item() for k, o, n, _ in slices if is_pre_bn_bias(k)), default=0.0)
    rep['pre_bn_bias_grad_max_over_wmax'] = zero_bias / wmax
    # (b) optimizer kernel vs torch.optim on the same gradient
    rule = _torch_rule(opt_kind, p_before, g_got, lr)
    step_size = (rule - p_before).abs().max().item()
    # both round p - step to fp32: allow 1 ulp of the parameter on top of 1e-4 of the step
    ulp = torch.finfo(torch.float32).eps * p_before.abs()
    rep['update_kernel_vs_torch_rule'] = ((p_after - rule).abs() - ulp).clamp_min(0).max().item() / max(step_size, 1e-30)
    # (b) applied update vs the oracle's, per tensor, over the elements whose gradient sign is settled
    # (|g_ref| above 20x the tensor's rms gradient error: RMSprop / Adam's first step is ~ lr * sign(g))
    p_ref_after = torch.cat([oracle_sd[k].detach().reshape(-1) for k, _, _, _ in slices])
    worst_u, worst_uk, settled = 0.0, None, 0
    for k, o, n, _ in keep:
        if norms[k] <= 1e-4 * nmax:
            continue
        gr, gg = oracle_grads[k].reshape(-1), g_got[o:o + n]
        rms_err = ((gg - gr).double().norm() / n ** 0.5).item()
        sel = gr.abs() > 20 * rms_err
        if int(sel.sum()) < 16:
            continue
        settled += int(sel.sum())
        du_got, du_ref = (p_after[o:o + n] - p_before[o:o + n])[sel], (p_ref_after[o:o + n] - p_before[o:o + n])[sel]
        e = rl2(du_got, du_ref)
        if e > worst_u:
            worst_u, worst_uk = e, k
    rep['worst_update_rel_l2'], rep['worst_update_tensor'] = worst_u, worst_uk
    rep['settled_fraction'] = settled / float(sum(n for _, _, n, _ in keep))
    rep['max_weight_diff_over_step'] = (p_after - p_ref_after).abs().max().item() / max(step_size, 1e-30)
    # (c) BN running statistics
    worst_bn = 0.0
    for k, v in net.state_dict().items():
        if 'running_' in k:
            ref = oracle_sd[k].detach()
            worst_bn = max(worst_bn, ((v.cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item())
        elif 'num_batches' in k:
            assert int(v) == int(oracle_sd[k]), (tag, which, k, int(v), int(oracle_sd[k]))
    rep['bn_running_rel_err'] = worst_bn
    _REPORT.setdefault(tag, {})[which] = rep
    _dump_report()
    print('\n[parity %s %s] %s' % (tag, which, json.dumps(rep)))
    lim = dict(LIMITS)
    lim.update(limits or {})
    return ['%s %s: %s = %.3g > %.3g' % (tag, which, k, rep[k], v) for k, v in lim.items() if rep[k] > v] + \
           (['%s %s: settled_fraction %.3f < 0.5' % (tag, which, rep['settled_fraction'])] if rep['settled_fraction'] < 0.5 else [])


def _dump_report():
    try:
        os.makedirs('gpurun_out', exist_ok=True)
        with open(os.path.join('gpurun_out', 'parity_fullsize_bwd.json'), 'w') as f:
            json.dump(_REPORT, f, indent=1, sort_keys=True)
    except OSError:
        pass


def _crit(p, name, C, per_band, sdV):
    cls = getattr(p.Loss, name)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        crit = cls(channel=C, perception_layer=1, perception_perBand=per_band, allow_seeded=True)
    crit.loss_perception.net.load_state_dict(sdV)
    return crit.to(DEV)


@pytest.mark.parametrize('literal', [False, True], ids=['minimal', 'literal'])
def test_rsss_iteration_gradients_full_size(conv_path, literal):
    """configs[2]: Demo_RSSS.py:285-332 at 13 bands 256x256, 2 tile pairs (S and D stepped, RMSprop 5e-5).  ``literal``: the
    reference's own call order -- Discriminator step through the un-detached change map with retain_graph, second Discriminator
    forward, Generator forward with a graph (steps.py) -- against the same oracle gradients (the literal order was only
    compared at fixture size)."""
    if literal and conv_path != 'winograd':
        pytest.skip('the literal order is checked on the default plan')
    p = pkg()
    C, N, H = 13, 2, 256
    sdG = seeded_state(onets.generator_spec(C), 11)
    sdS = seeded_state(onets.segmentor_spec(C, 1, True), 12)
    sdD = seeded_state(onets.discriminator_spec(C), 13)
    sdV = seeded_state(onets.vgg_spec(), 4242)
    x, y, region = seeded_tiles(21, N, C, H, H)
    if 'rsss' not in _ORACLE:
        n = osteps.Nets(sdG, sdS, sdD, sdV).make_optimizers('rsss')
        n.capture = {}
        ro = osteps.rsss_adversarial_step(n, x, y, region)
        g64 = _oracle_fp64((sdG, sdS, sdD, sdV), 'rsss',
                           lambda m: osteps.rsss_adversarial_step(m, _t64(x), _t64(y), _t64(region)))
        _ORACLE['rsss'] = (n, ro, g64, _truth_limits(n.capture, g64, 'D', True), _truth_limits(n.capture, g64, 'S'), {})
    n, ro, g64, limD, limS, dcache = _ORACLE['rsss']

    def d_truth(cm):                      # Demo_RSSS.py:288-300 with the given density map (discriminator_continuous)
        xd, yd, rd = x.double(), y.double(), region.double()
        keep = 1 - cm.detach().cpu().double()
        return _d_step_fp64(sdD, (xd * keep, yd * keep), (xd * keep, (yd * (1 - rd) + xd * rd) * keep))
    netG, netS, netD = p.Module.Generator(C), p.Module.Segmentor(C, 1, True), p.Module.Discriminator_SRGAN_simple(C)
    netG.load_state_dict(sdG); netS.load_state_dict(sdS); netD.load_state_dict(sdD)
    crit = _crit(p, 'CGeneratorLoss', C, True, sdV)
    for m in (netG, netS, netD):
        m.to(DEV)
    netS.train(); netD.train(); netG.eval()
    oS, oD = p.optim.RMSprop(netS.parameters(), lr=5e-5), p.optim.RMSprop(netD.parameters(), lr=5e-5)
    store = {}
    oS.pre_step_hooks.append(_hook(store, 'S'))
    oD.pre_step_hooks.append(_hook(store, 'D'))
    r = p.steps.rsss_adversarial_step(netS, netD, netG, crit, oS, oD, x.to(DEV), y.to(DEV), region.to(DEV), literal=literal)
    # values: every logged loss, the change-density map (1e-4, north_star) and the thresholded map
    got = [float(r[k]) for k in ('d_loss', 's_loss', 's_d_loss', 'g_loss', 'l1_loss', 'r_loss', 'generator_loss', 'ssim_loss', 'perception_loss')]
    ref = [float(ro[k]) for k in ('d_loss', 's_loss', 's_d_loss', 'g_loss', 'l1_loss', 'r_loss', 'gen', 'ssim', 'perc')]
    np.testing.assert_allclose(got, ref, rtol=5e-4, atol=1e-6)
    cm, cmo = r['cmap'].detach().cpu(), ro['cmap'].detach()
    err = (cm - cmo).abs().max().item()
    assert err <= 1e-4, 'density map L_inf %.2e' % err
    safe = (cmo - 0.5).abs() > 2e-4
    assert torch.equal((cm > 0.5)[safe], (cmo > 0.5)[safe])
    tag = 'rsss_13x256_' + conv_path + ('_literal' if literal else '')
    own = _d_own_truths(d_truth, g64, r['cmap'].detach().cpu(), ro['cmap'].detach(), dcache)
    bad = check_net(tag, 'D', netD, 'rmsprop', 5e-5, store, n.capture['D'], n.D, g64['D'], conv_path, limD, truth_own=own)
    bad += check_net(tag, 'S', netS, 'rmsprop', 5e-5, store, n.capture['S'], n.S, g64['S'], conv_path, limS)
    assert not bad, bad


def test_usss_generator_step_gradients_full_size(conv_path):
    """configs[1]: Demo_USSS.py:142-159 at 4 bands 256x256, 2 tiles (G stepped, Adam 2e-4, train-mode BN)."""
    p = pkg()
    C, N, H = 4, 2, 256
    sdG = seeded_state(onets.generator_spec(C), 41)
    sdV = seeded_state(onets.vgg_spec(), 4242)
    x, y, _ = seeded_tiles(43, N, C, H, H)
    if 'usss' not in _ORACLE:
        n = osteps.Nets(sdG, None, None, sdV)
        n.opt['G'] = torch.optim.Adam(n.params('G'), lr=2e-4, betas=(0.9, 0.99))
        n.capture = {}
        ro = osteps.usss_g_pretrain_step(n, x, y)

        def run64(m):
            m.opt['G'] = torch.optim.Adam(m.params('G'), lr=2e-4, betas=(0.9, 0.99))
            osteps.usss_g_pretrain_step(m, _t64(x), _t64(y))
        g64 = _oracle_fp64((sdG, None, None, sdV), None, run64)
        _ORACLE['usss'] = (n, ro, g64)
    n, ro, g64 = _ORACLE['usss']
    netG = p.Module.Generator(C)
    netG.load_state_dict(sdG)
    crit = _crit(p, 'CNetLoss', C, True, sdV)
    netG.to(DEV).train()
    oG = p.optim.Adam(netG.parameters(), lr=2e-4, betas=(0.9, 0.99))
    store = {}
    oG.pre_step_hooks.append(_hook(store, 'G'))
    r = p.steps.usss_g_pretrain_step(netG, crit, oG, x.to(DEV), y.to(DEV))
    np.testing.assert_allclose([float(r['loss']), float(r['generator_loss']), float(r['perception_loss']), float(r['ssim_loss'])],
                               [float(ro['loss']), float(ro['gen']), float(ro['perc']), float(ro['ssim'])], rtol=5e-4, atol=1e-6)
    bad = check_net('usss_g_4x256_' + conv_path, 'G', netG, 'adam', 2e-4, store, n.capture['G'], n.G, g64['G'], conv_path)
    assert not bad, bad


def test_usss_generator_gradient_with_direct_vgg_decisions():
    """the same Generator step on the Winograd plan -- F(4x4) values and arithmetic in every wide layer, forward
    and backward -- but with the ReLU mask / max-pool code of every frozen VGG layer taken from the direct kernels' forward pass
    (tests/decisions.py).  The flat gradient then lands within 2x of stock fp32's own distance to the fp64 gradient (measured 1.12x;
    5.42x with the F(4x4) forward's own decisions): the F(4x4) ARITHMETIC is as accurate as the reference's, what the end-to-end
    figure measures is how many activation decisions sit within F(4x4) rounding of their kink."""
    from decisions import inject_direct_vgg_decisions
    from fcd_gan_pytorch_amd import _lib
    p = pkg()
    C, N, H = 4, 2, 256
    sdG = seeded_state(onets.generator_spec(C), 41)
    sdV = seeded_state(onets.vgg_spec(), 4242)
    x, y, _ = seeded_tiles(43, N, C, H, H)
    if 'usss' not in _ORACLE:
        n = osteps.Nets(sdG, None, None, sdV)
        n.opt['G'] = torch.optim.Adam(n.params('G'), lr=2e-4, betas=(0.9, 0.99))
        n.capture = {}
        ro = osteps.usss_g_pretrain_step(n, x, y)

        def run64(m):
            m.opt['G'] = torch.optim.Adam(m.params('G'), lr=2e-4, betas=(0.9, 0.99))
            osteps.usss_g_pretrain_step(m, _t64(x), _t64(y))
        _ORACLE['usss'] = (n, ro, _oracle_fp64((sdG, None, None, sdV), None, run64))
    n, ro, g64 = _ORACLE['usss']
    names = [k for k in g64['G'] if g64['G'][k] is not None and not is_pre_bn_bias(k)]
    flat64 = torch.cat([g64['G'][k].reshape(-1) for k in names])
    err32 = ((torch.cat([n.capture['G'][k].double().reshape(-1) for k in names]) - flat64).norm() / flat64.norm()).item()

    def hip_error(inject):
        netG = p.Module.Generator(C)
        netG.load_state_dict(sdG)
        crit = _crit(p, 'CNetLoss', C, True, sdV)
        netG.to(DEV).train()
        import contextlib
        with (inject_direct_vgg_decisions() if inject else contextlib.nullcontext()):
            y_fake = netG(x.to(DEV))
            gen, l1, perc, ssim = crit(y.to(DEV), y_fake, torch.zeros((N, 1, H, H), device=DEV))
            (gen + 0.4 * perc).backward()
        got = dict(netG.named_parameters())
        flat = torch.cat([got[k].grad.detach().cpu().double().reshape(-1) for k in names])
        return ((flat - flat64).norm() / flat64.norm()).item()
    prev = _lib.lib.fcd_conv_wino_set(4)
    try:
        injected, own = hip_error(True), hip_error(False)
    finally:
        _lib.lib.fcd_conv_wino_set(prev)
    _REPORT['usss_g_4x256_decisions'] = dict(oracle32_vs_fp64=err32, winograd_own_decisions=own, winograd_direct_vgg_decisions=injected,
                                             ratio_own=own / err32, ratio_injected=injected / err32)
    _dump_report()
    assert injected <= 2.0 * err32 + 2e-4, (injected, err32)
    assert injected < own                       # the decisions are where the distance comes from


def test_wsss_iteration_gradients_full_size(conv_path):
    """configs[4]: Demo_WSSS.py:249-323 at 3 bands 512x512, one changed + one unchanged pair
    (S RMSprop 1e-3, D RMSprop 1e-5)."""
    p = pkg()
    C, N, H = 3, 1, 512
    sdG = seeded_state(onets.generator_spec(C), 51)
    sdS = seeded_state(onets.segmentor_spec(C, 1, True), 52)
    sdD = seeded_state(onets.discriminator_spec(C), 53)
    sdV = seeded_state(onets.vgg_spec(), 4242)
    x, y, _ = seeded_tiles(54, N, C, H, H)
    xn, yn, _ = seeded_tiles(55, N, C, H, H)
    yn = xn + 0.1 * (yn - xn)
    if 'wsss' not in _ORACLE:
        n = osteps.Nets(sdG, sdS, sdD, sdV).make_optimizers('wsss')
        n.capture = {}
        ro = osteps.wsss_adversarial_step(n, x, y, xn, yn)
        g64 = _oracle_fp64((sdG, sdS, sdD, sdV), 'wsss',
                           lambda m: osteps.wsss_adversarial_step(m, _t64(x), _t64(y), _t64(xn), _t64(yn)))
        _ORACLE['wsss'] = (n, ro, g64, _truth_limits(n.capture, g64, 'D', True), _truth_limits(n.capture, g64, 'S'), {})
    n, ro, g64, limD, limS, dcache = _ORACLE['wsss']

    def d_truth(cm):                      # Demo_WSSS.py:262-285: both pairs masked with the CHANGED pair's map
        keep = 1 - cm.detach().cpu().double()
        return _d_step_fp64(sdD, (x.double() * keep, y.double() * keep), (xn.double() * keep, yn.double() * keep))
    netG, netS, netD = p.Module.Generator(C), p.Module.Segmentor(C, 1, True), p.Module.Discriminator_SRGAN_simple(C)
    netG.load_state_dict(sdG); netS.load_state_dict(sdS); netD.load_state_dict(sdD)
    crit = _crit(p, 'CGeneratorLoss', C, False, sdV)
    for m in (netG, netS, netD):
        m.to(DEV)
    netS.train(); netD.train(); netG.eval()
    oS, oD = p.optim.RMSprop(netS.parameters(), lr=1e-3), p.optim.RMSprop(netD.parameters(), lr=1e-5)
    store = {}
    oS.pre_step_hooks.append(_hook(store, 'S'))
    oD.pre_step_hooks.append(_hook(store, 'D'))
    r = p.steps.wsss_adversarial_step(netS, netD, netG, crit, oS, oD, x.to(DEV), y.to(DEV), xn.to(DEV), yn.to(DEV))
    got = [float(r[k]) for k in ('d_loss', 's_loss', 's_d_loss', 'g_loss', 'l1_loss', 'nc_loss', 'generator_loss', 'ssim_loss', 'perception_loss')]
    ref = [float(ro[k]) for k in ('d_loss', 's_loss', 's_d_loss', 'g_loss', 'l1_loss', 'nc_loss', 'gen', 'ssim', 'perc')]
    np.testing.assert_allclose(got, ref, rtol=5e-4, atol=1e-6)
    for a, b in ((r['cmap'], ro['cmap']), (r['ncmap'], ro['ncmap'])):
        assert (a.detach().cpu() - b.detach()).abs().max().item() <= 1e-4
    tag = 'wsss_3x512_' + conv_path
    own = _d_own_truths(d_truth, g64, r['cmap'].detach().cpu(), ro['cmap'].detach(), dcache)
    bad = check_net(tag, 'D', netD, 'rmsprop', 1e-5, store, n.capture['D'], n.D, g64['D'], conv_path, limD, truth_own=own)
    bad += check_net(tag, 'S', netS, 'rmsprop', 1e-3, store, n.capture['S'], n.S, g64['S'], conv_path, limS)
    assert not bad, bad


def test_discriminator_step_gradient_error_distribution():
    """the Discriminator's gradient accuracy settled as a DISTRIBUTION instead of one absolute bound.
    On 16 density maps differing by <= 3e-5 (both HIP plans' maps, seeded 1e-5 noise draws, constant offsets) the Demo_RSSS D-step
    gradient (13 bands 256 x 256, 2 pairs) is evaluated in fp64 (the truth for that map), on the fp32 CPU oracle and on the HIP
    kernels.  Each fp32 evaluation is either at rounding level (~3e-6) or one activation decision away (2e-4 ... 1e-2); the rule
    compares the two implementations where that is meaningful:
      floor   -- the best HIP draw is as close to fp64 as the best oracle draw (<= 2x): the arithmetic itself is as accurate;
      median, geometric mean, max over the maps <= 2x / 4x / 2x the oracle's;
      jumps   -- HIP takes a discrete jump (error > 1e-4) on at most 4 maps more than the oracle does.
    Numbers: profiles/r04_parity_d_probe.md (same protocol, tools/parity_probe_d.py, incl. the pooled-vs-difference A/B: the
    order of AdaptiveAvgPool and the pair difference does not move any of the 16 errors in the third digit)."""
    p = pkg()
    from fcd_gan_pytorch_amd import _lib
    C, N, H = 13, 2, 256
    sdD = seeded_state(onets.discriminator_spec(C), 13)
    sdS = seeded_state(onets.segmentor_spec(C, 1, True), 12)
    xc, yc, rc = seeded_tiles(21, N, C, H, H)
    x, y, region = xc.to(DEV), yc.to(DEV), rc.to(DEV)
    cms = []
    prev = _lib.lib.fcd_conv_wino_set(-1)
    try:
        for plan in (0, 4):
            _lib.lib.fcd_conv_wino_set(plan)
            S = p.Module.Segmentor(C, 1, True); S.load_state_dict(sdS); S.to(DEV).train()
            with torch.no_grad():
                cms.append(S(x, y))
    finally:
        _lib.lib.fcd_conv_wino_set(prev)
    i = 0
    while len(cms) < 16:
        g = torch.Generator(device=DEV).manual_seed(100 + i)
        if i % 4 == 3:
            cms.append(cms[0] + (i // 4 + 1) * 1e-5 * (-1) ** (i // 4))
        else:
            cms.append(cms[0] + 1e-5 * torch.randn(cms[0].shape, device=DEV, generator=g))
        i += 1
    eh, eo = [], []
    for cm in cms:
        keep64 = 1 - cm.detach().cpu().double()
        xd, yd, rd = xc.double(), yc.double(), rc.double()
        t = _d_step_fp64(sdD, (xd * keep64, yd * keep64), (xd * keep64, (yd * (1 - rd) + xd * rd) * keep64))
        oD = onets.clone_state(sdD)
        keep32 = 1 - cm.detach().cpu()
        c = onets.discriminator(oD, xc * keep32, yc * keep32, train=True)
        nc = onets.discriminator(oD, xc * keep32, (yc * (1 - rc) + xc * rc) * keep32, train=True)
        (1 + nc.mean() - c.mean()).backward()
        D = p.Module.Discriminator_SRGAN_simple(C); D.load_state_dict(sdD); D.to(DEV).train()
        keep = 1 - cm.detach()
        c_out, nc_out = D.forward_pairs([(x * keep, y * keep), (x * keep, (y * (1 - region) + x * region) * keep)])
        (1 + nc_out.mean() - c_out.mean()).backward()
        ks = [k for k in t if not is_pre_bn_bias(k)]
        gh = dict(D.named_parameters())
        ft = _flat64(t, ks)
        eh.append(((torch.cat([gh[k].grad.detach().cpu().double().reshape(-1) for k in ks]) - ft).norm() / ft.norm()).item())
        eo.append(((torch.cat([oD[k].grad.detach().double().reshape(-1) for k in ks]) - ft).norm() / ft.norm()).item())
    eh, eo = np.array(eh), np.array(eo)
    rep = dict(hip=eh.tolist(), oracle32=eo.tolist(), hip_median=float(np.median(eh)), oracle32_median=float(np.median(eo)),
               hip_max=float(eh.max()), oracle32_max=float(eo.max()), hip_min=float(eh.min()), oracle32_min=float(eo.min()),
               hip_geomean=float(np.exp(np.log(eh).mean())), oracle32_geomean=float(np.exp(np.log(eo).mean())),
               hip_jumps=int((eh > 1e-4).sum()), oracle32_jumps=int((eo > 1e-4).sum()))
    _REPORT['d_step_error_distribution_16_maps'] = rep
    _dump_report()
    print('\n[D distribution] %s' % json.dumps({k: v for k, v in rep.items() if not isinstance(v, list)}))
    assert rep['hip_min'] <= 2 * rep['oracle32_min'] + 1e-6, rep
    assert rep['hip_median'] <= 2 * rep['oracle32_median'] + 1e-5, rep
    assert rep['hip_geomean'] <= 4 * rep['oracle32_geomean'], rep
    assert rep['hip_max'] <= 2 * rep['oracle32_max'], rep
    assert rep['hip_jumps'] <= rep['oracle32_jumps'] + 4, rep
