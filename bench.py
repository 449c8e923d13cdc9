#!/usr/bin/env python3
"""Headline benchmark: tile-pairs/s of the FCD-GAN RSSS adversarial train step
(Demo_RSSS.py:285-332: S fwd, D step, S step incl. eval-mode G, masked MSE,
MS-SSIM, per-band VGG16 perception, region losses, RMSprop) on 13-band 256x256
synthetic bi-temporal tiles, fp32, one process per GPU (weak scaling: fixed tiles
per GPU, gradients all-reduced over RCCL).

    python bench.py --gpus N --steps K --warmup W
    N > 1 without a torchrun environment: bench.py launches the N ranks itself (re-exec through
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`, one rank per GPU,
    backend nccl = RCCL); under torchrun (RANK / LOCAL_RANK / WORLD_SIZE set by the launcher) it is one of the ranks.

Prints ONE JSON line on rank 0 (see the driver contract) with two extra objects:
  roofline     -- the MFMA kernel family with the largest share of the step (today: the batched
                  GEMMs of the Winograd F(4x4,3x3) layers), timed with HIP events on the launch
                  stream in a second, profiled pass: executed FLOPs / event time against the peak
                  of the matrix pipe the family REALLY runs on -- `pipe` says which: "bf16x6" =
                  v_mfma_f32_32x32x16_bf16 on exact three-way operand splits, six bf16 products per
                  fp32 multiply, priced as 6 x the fp32-equivalent FLOPs over the dense bf16 peak
                  (2500 TFLOP/s); "fp32" = v_mfma_f32_*, priced over 157.3 TFLOP/s
                  (MI355X_MICROARCH.md).  `other_mfma_kernels` lists the other families the same way.
  cpu_baseline -- the CPU oracle (port of the reference's step on stock torch CPU ops)
                  timed on this box's host cores on a bounded sample (rank 0, N=1 only)
`config.arithmetic` is generated from the library's switch state (fcd_conv_wino_split_set /
fcd_conv_wgrad_split_set), and `fp32_mfma_only` is the same step with BOTH of them off: every
matrix instruction of the step on the fp32 pipe.
"""
import argparse
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense bf16 MFMA (MI355X_MICROARCH.md); the split GEMM executes 6 bf16 FLOPs per fp32-equivalent FLOP


WORKLOADS = {
    # name: (BASELINE.json config, default bands, size, tile pairs per GPU, description)
    'rsss': (2, 13, 256, 8, 'Demo_RSSS adversarial step (S+D+G, masked MSE, MS-SSIM, per-band VGG16 perception, '
                            'region losses, RMSprop)'),
    'usss_g': (1, 4, 256, 16, 'Demo_USSS generator-only pre-training step (G fwd/bwd, masked L1 with cmap=0, MS-SSIM, '
                              'per-band VGG16 perception, Adam)'),
    'wsss': (4, 3, 512, 4, 'Demo_WSSS adversarial step (S on the changed and the unchanged pair, D, eval-mode G, masked '
                           'MSE, MS-SSIM, RGB VGG16 perception, RMSprop)'),
}


def arithmetic_text(wino_split, wgrad_split):
    """What the step computes in, from the state of the two pipe switches (never a hand-written claim)."""
    base = 'fp32 tensors, fp32 accumulation everywhere.  Matrix pipes: '
    split_how = ('v_mfma_f32_32x32x16_bf16 with every fp32 operand split EXACTLY into three bf16 parts and six partial products per '
                 'multiply summed in fp32 (dropped terms <= 2^-24 relative: fp32-equivalent, measured error vs fp64 <= that of the fp32 '
                 'MFMA path, tests/test_gpu_ops.py, tests/test_split_arithmetic.py)')
    on_split, on_fp32 = [], ['the direct implicit-GEMM convolutions (forward / data gradient)', 'the fused F(2x2,3x3) kernel',
                             'the thin-channel / 9x9 / 1x1 weight-gradient kernels']
    (on_split if wino_split else on_fp32).append('the batched GEMMs of the Winograd F(4x4,3x3) layers (forward, data gradient, weight gradient)')
    (on_split if wgrad_split else on_fp32).append('the NCHW-direct 3x3 weight-gradient kernel of the layers off the F(4x4) form '
                                                  '(stride 2: on the split pipe regardless)' if wgrad_split else
                                                  'the NCHW-direct 3x3 weight-gradient kernel (stride-1 layers; the stride-2 ones fall back '
                                                  'to channel-minor copies + the fp32 kernel)')
    txt = base + 'v_mfma_f32_* (fp32 matrix pipe) for ' + '; '.join(on_fp32) + '.'
    if on_split:
        txt += '  ' + split_how + ' for ' + '; '.join(on_split) + '.  `fp32_mfma_only` below is the same step with every one of those on the fp32 pipe.'
    return txt


def mfma_entry(name, pipe, ms, fp32_equiv_flops, launches, what, psteps, dt_prof):
    """One roofline entry.  ``pipe`` 'fp32': executed FLOPs = the FLOPs given, peak 157.3; 'bf16x6': the kernel executes SIX bf16
    MFMA FLOPs per fp32-equivalent FLOP and is priced on those against the dense bf16 peak."""
    mult, peak = (6.0, PEAK_BF16_MFMA_TFLOPS) if pipe == 'bf16x6' else (1.0, PEAK_F32_MFMA_TFLOPS)
    flops = mult * fp32_equiv_flops
    ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    e = {'bound': 'mfma', 'kernel': name, 'pipe': pipe, 'flops_counted': what, 'achieved': ach,
         'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak,
         'traffic': None, 'launches_per_step': launches / psteps,
         'avg_launch_ms': ms / max(launches, 1), 'gflop_per_launch': flops / max(launches, 1) / 1e9,
         'share_of_step_time': ms / (1e3 * dt_prof) if dt_prof else None}
    if pipe == 'bf16x6':
        e['fp32_equivalent_tflops'] = ach / 6.0
        e['fp32_equivalent_over_fp32_mfma_peak'] = ach / 6.0 / PEAK_F32_MFMA_TFLOPS
        e['gflop_per_launch_fp32_equivalent'] = e['gflop_per_launch'] / 6.0
    return e


def roofline_entries(prof, psteps, dt_prof):
    """The MFMA-bound kernel families of the profiled pass, each priced on the FLOPs its launches really issue against the peak of the
    pipe they really run on, sorted by share of the step:
      * conv_igemm*: direct implicit-GEMM convolution -> algorithmic FLOPs 2 N K P Q C R S, fp32 pipe
      * wino_gemm: the batched GEMM of the Winograd path on the fp32 pipe -> 2 (m+2)^2 rows Kc T
      * wino_gemm_bf16x6: the same GEMMs on the bf16 pipe (exact split) -> 6 x that
      * conv_wgrad minus its nested Winograd and bf16-split parts: the fp32-pipe weight-gradient kernels (re-layout + reduce included)
      * conv_wgrad_bf16x6: the NCHW-direct 3x3 weight-gradient kernel on the bf16 pipe -> 6 x the algorithmic weight-gradient FLOPs
      * conv_wino2*: fused F(2x2,3x3), fp32 pipe, executed = algorithmic x 16/36"""
    zero = dict(ms=0.0, launches=0, flops=0.0, bytes=0.0)
    g = lambda k: prof.get(k, zero)
    fwd, dg, wg = g('conv_igemm_fwd'), g('conv_igemm_dgrad'), g('conv_wgrad')
    w2f, w2d = g('conv_wino2_fwd'), g('conv_wino2_dgrad')
    wgemm, wsplit, wgw, wgs = g('wino_gemm'), g('wino_gemm_bf16x6'), g('conv_wgrad_wino'), g('conv_wgrad_bf16x6')
    E = lambda *a: mfma_entry(*a, psteps, dt_prof)
    cands = [
        E('conv_igemm_kernel / conv_igemm_glds_kernel (direct fp32 MFMA implicit-GEMM convolution; forward + data-gradient launches)',
          'fp32', fwd['ms'] + dg['ms'], fwd['flops'] + dg['flops'], fwd['launches'] + dg['launches'], 'algorithmic convolution FLOPs'),
        E('wino_gemm_kernel (batched fp32 MFMA GEMM of the Winograd F(4x4,3x3) path; forward, data-gradient and weight-gradient launches)',
          'fp32', wgemm['ms'], wgemm['flops'], wgemm['launches'], 'executed GEMM FLOPs'),
        E('weight gradient, fp32-pipe kernels: conv_wgrad_roll_kernel / conv_wgrad_kernel / thin-channel and 9x9 kernels (+ channel-minor '
          're-layout, split-K reduce).  The wide 3x3 layers take the Winograd F(4x4,3x3) form (priced under wino_gemm*, transforms under '
          'kernel_families.wino_transform) and the NCHW-direct 3x3 kernel runs on the bf16 pipe (own entry): neither is in here',
          'fp32', wg['ms'] - wgw['ms'] - wgs['ms'], wg['flops'] - wgw['flops'] - wgs['flops'],
          wg['launches'] - wgw['launches'] - wgs['launches'], 'executed = algorithmic weight-gradient FLOPs of the layers on these kernels'),
        E('wino_gemm_split256_kernel / wino_gemm_split_kernel / wino_gemm_split_res_kernel (batched GEMM of the Winograd F(4x4,3x3) path, '
          'forward, data-gradient and weight-gradient launches, on the bf16 matrix pipe: every fp32 operand split exactly into three bf16 '
          'parts, six partial products per multiply accumulated in fp32 -- fp32-equivalent results, tests/test_gpu_ops.py)',
          'bf16x6', wsplit['ms'], wsplit['flops'], wsplit['launches'], 'executed bf16 MFMA FLOPs = 6 x the fp32-equivalent GEMM FLOPs'),
        E('conv_wgrad_roll_nchw_kernel<split> (NCHW-direct 3x3 weight gradient, stride 1 and 2, on the bf16 matrix pipe: x and dY split '
          'exactly into three bf16 parts in registers, six products per multiply accumulated in fp32)',
          'bf16x6', wgs['ms'], wgs['flops'], wgs['launches'], 'executed bf16 MFMA FLOPs = 6 x the algorithmic weight-gradient FLOPs'),
        E('conv_wino2_kernel (fused Winograd F(2x2,3x3): input transform + sixteen 16x16x4 fp32 MFMA GEMMs + output transform in one '
          'kernel; the 64-row 3x3 layers, forward + data gradient)', 'fp32', w2f['ms'] + w2d['ms'],
          (w2f['flops'] + w2d['flops']) * 16.0 / 36.0, w2f['launches'] + w2d['launches'],
          'executed MFMA FLOPs (= algorithmic conv FLOPs x 16/36)'),
    ]
    cands = [e for e in cands if e['launches_per_step'] > 0]
    cands.sort(key=lambda e: -(e['share_of_step_time'] or 0.0))
    return cands


def check_result_consistency(res):
    """Self-consistency of an emitted bench line (tests/test_bench_launch.py runs it over the committed lines; main() over the
    line it is about to print): every roofline entry is priced against the peak of the pipe it names."""
    errs = []
    rl = res.get('roofline')
    for e in ([rl] + list(rl.get('other_mfma_kernels', []))) if rl else []:
        peak = PEAK_BF16_MFMA_TFLOPS if e.get('pipe') == 'bf16x6' else PEAK_F32_MFMA_TFLOPS
        if e.get('pipe') not in ('fp32', 'bf16x6'):
            errs.append('%s: no pipe' % e['kernel'][:40])
        if abs(e['peak'] - peak) > 1e-9:
            errs.append('%s: pipe %s priced against %s' % (e['kernel'][:40], e.get('pipe'), e['peak']))
        if abs(e['frac'] - e['achieved'] / e['peak']) > 1e-9 * max(1.0, e['frac']):
            errs.append('%s: frac != achieved / peak' % e['kernel'][:40])
        if e.get('pipe') == 'bf16x6' and abs(e['fp32_equivalent_tflops'] * 6.0 - e['achieved']) > 1e-6 * e['achieved']:
            errs.append('%s: fp32-equivalent figure is not achieved / 6' % e['kernel'][:40])
        if ('split' in e['kernel'].split('(')[0]) != (e.get('pipe') == 'bf16x6'):      # (the kernel names before the description)
            errs.append('%s: kernel name and pipe disagree' % e['kernel'][:40])
    sw = (res.get('config') or {}).get('pipes')
    if sw is not None:
        txt = res['config']['arithmetic']
        if txt != arithmetic_text(sw['wino_split'], sw['wgrad_split']):
            errs.append('config.arithmetic does not describe config.pipes')
    if 'value' in res and 'ms_per_step' in res and res.get('config', {}).get('global_batch'):
        v = res['config']['global_batch'] / (res['ms_per_step'] * 1e-3)
        if abs(v - res['value']) > 1e-6 * v:
            errs.append('value != global_batch / ms_per_step')
    return errs


def build_workload(args, dev, rank):
    import fcd_gan_pytorch_amd as fcd
    from fcd_gan_pytorch_amd.synthetic import synthetic_tiles
    C, H, W, N = args.bands, args.size, args.size, args.batch
    torch.manual_seed(0)
    netD = fcd.Module.Discriminator_SRGAN_simple(n_channels=C)
    netS = fcd.Module.Segmentor(n_channels=C, bilinear=True)
    netG = fcd.Module.Generator(n_channels=C)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        if args.workload == 'usss_g':
            crit = fcd.Loss.CNetLoss(channel=C, perception_layer=1, perception_perBand=True, allow_seeded=True)      # Demo_USSS.py:116
        else:
            crit = fcd.Loss.CGeneratorLoss(channel=C, perception_layer=1, perception_perBand=args.workload == 'rsss', allow_seeded=True)
    for m in (netD, netS, netG, crit):
        m.to(dev)
    multi = dist.is_initialized()          # (a forced one-rank group runs the broadcasts too)
    x, y, region = (t.to(dev) for t in synthetic_tiles(1234 + rank, N, C, H, W))

    if args.workload == 'usss_g':
        netG.train()                                      # Demo_USSS.py:127
        optG = fcd.optim.Adam(netG.parameters(), lr=1e-4, betas=(0.9, 0.99))
        if multi:
            dist.broadcast(optG.flat_p, 0)
            for t in netG.buffers():
                dist.broadcast(t, 0)

        def step():
            return fcd.steps.usss_g_pretrain_step(netG, crit, optG, x, y)
        return step, {'G': optG}

    netS.train(); netD.train(); netG.eval()          # Demo_RSSS.py:146-148,240 / Demo_WSSS.py:206
    optS = fcd.optim.RMSprop(netS.parameters(), lr=5e-5)
    optD = fcd.optim.RMSprop(netD.parameters(), lr=5e-5)
    if multi:
        dist.broadcast(optS.flat_p, 0)
        dist.broadcast(optD.flat_p, 0)
        for t in list(netG.state_dict().values()) + list(netS.buffers()) + list(netD.buffers()):
            dist.broadcast(t, 0)
    if args.workload == 'wsss':
        # the unchanged pair: T2 = T1 + small noise, no changed rectangle (WHU_Dataset_WSS pairs a
        # changed sample with an unchanged one, data_utils.py:570-625)
        x_nc, _, _ = synthetic_tiles(4321 + rank, N, C, H, W)
        x_nc = x_nc.to(dev)
        g = torch.Generator(device=dev).manual_seed(99 + rank)
        y_nc = x_nc + 0.1 * torch.randn(x_nc.shape, device=dev, generator=g)

        def step():
            return fcd.steps.wsss_adversarial_step(netS, netD, netG, crit, optS, optD, x, y, x_nc, y_nc)
        return step, {'S': optS, 'D': optD}

    def step():
        return fcd.steps.rsss_adversarial_step(netS, netD, netG, crit, optS, optD, x, y, region)
    return step, {'S': optS, 'D': optD}


def write_layer_tables(path, detail, psteps, args):
    """Per-layer tables of the profiled pass (one row per distinct launch geometry): avg launch time, achieved
    TFLOP/s (or GB/s) against the fp32 MFMA peak, and for the Winograd GEMM the workgroup count / chip fill."""
    import re
    from collections import OrderedDict
    groups = OrderedDict()
    for e in detail:
        g = groups.setdefault((e['family'], e['tag']), dict(n=0, ms=0.0, flops=0.0, bytes=0.0))
        g['n'] += 1; g['ms'] += e['ms']; g['flops'] += e['flops']; g['bytes'] += e['bytes']

    def rows(fam):
        out = [(tag, g) for (f, tag), g in groups.items() if f == fam]
        out.sort(key=lambda r: -r[1]['ms'])
        return out

    L = ['# Per-layer launch table (bench.py --layers-md, %s workload, %d tile pairs/GPU, %d bands %dx%d, %d profiled step(s))'
         % (args.workload, args.batch, args.bands, args.size, args.size, psteps), '',
         'HIP events around every launch on its stream; fp32 MFMA peak %.1f TFLOP/s.  "/step" = launches per step.' % PEAK_F32_MFMA_TFLOPS, '']
    L += ['## wino_gemm_kernel (executed GEMM FLOPs = 2 x batch x M x N x Kc)', '',
          '| launch | /step | avg us | ms/step | GFLOP | TFLOP/s | of peak | workgroups | resident slots filled |',
          '|---|---|---|---|---|---|---|---|---|']
    tot_ms = tot_fl = 0.0
    for tag, g in rows('wino_gemm'):
        m = re.search(r'M=(\d+) N=(\d+) Kc=(\d+) batch=(\d+)(?: splits=(\d+))?', tag)
        M, N, Kc, B = (int(v) for v in m.groups()[:4])
        splits = int(m.group(5) or 1)
        if M <= 64:
            wgs, per_cu = ((N + 255) // 256) * B * splits, 2
        else:
            wgs, per_cu = ((M + 127) // 128) * ((N + 127) // 128) * B * splits, 2
        waves = wgs / float(256 * per_cu)
        fill = wgs / (256.0 * per_cu * max(1, -(-wgs // (256 * per_cu))))       # average occupancy of the slot waves
        us = 1e3 * g['ms'] / g['n']
        tf = g['flops'] / (g['ms'] * 1e-3) / 1e12 if g['ms'] > 0 else 0.0
        tot_ms += g['ms']; tot_fl += g['flops']
        L.append('| %s | %.1f | %.1f | %.3f | %.2f | %.1f | %.2f | %d | %.2f waves, %.0f%% |' % (
            tag, g['n'] / float(psteps), us, g['ms'] / psteps, g['flops'] / g['n'] / 1e9, tf, tf / PEAK_F32_MFMA_TFLOPS, wgs,
            waves, 100 * fill))
    if tot_ms > 0:
        L += ['', 'all launches: %.2f ms/step, %.1f TFLOP/s = %.3f of peak' % (
            tot_ms / psteps, tot_fl / (tot_ms * 1e-3) / 1e12, tot_fl / (tot_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS)]
    rr = rows('wino_gemm_bf16x6')
    if rr:
        L += ['', '## wino_gemm_split kernels: the same batched GEMM on the bf16 matrix pipe, fp32 operands split exactly into three '
              'bf16 parts, six partial products accumulated in fp32', '',
              'GFLOP = fp32-equivalent GEMM FLOPs (2 x batch x M x N x Kc); the MFMAs execute 6x that in bf16; bf16 dense peak %.0f TFLOP/s' % PEAK_BF16_MFMA_TFLOPS, '',
              '| launch | /step | avg us | ms/step | GFLOP (fp32-equiv) | TFLOP/s fp32-equiv | x of the fp32 MFMA peak | bf16 TFLOP/s executed | of bf16 peak | GB (V + M + filter planes, once) | TB/s on those bytes |',
              '|---|---|---|---|---|---|---|---|---|---|---|']
        tms = tfl = 0.0
        for tag, g in rr:
            us = 1e3 * g['ms'] / g['n']
            tf = g['flops'] / (g['ms'] * 1e-3) / 1e12 if g['ms'] > 0 else 0.0
            tms += g['ms']; tfl += g['flops']
            L.append('| %s | %.1f | %.1f | %.3f | %.2f | %.1f | %.2f | %.0f | %.2f | %.2f | %.2f |' % (
                tag, g['n'] / float(psteps), us, g['ms'] / psteps, g['flops'] / g['n'] / 1e9, tf, tf / PEAK_F32_MFMA_TFLOPS,
                6 * tf, 6 * tf / PEAK_BF16_MFMA_TFLOPS, g['bytes'] / g['n'] / 1e9, g['bytes'] / (g['ms'] * 1e-3) / 1e12 if g['ms'] > 0 else 0.0))
        tf = tfl / (tms * 1e-3) / 1e12
        L += ['', 'all launches: %.2f ms/step, %.1f TFLOP/s fp32-equivalent (%.2fx the fp32 MFMA peak), %.0f TFLOP/s bf16 executed = %.3f of the bf16 peak'
              % (tms / psteps, tf, tf / PEAK_F32_MFMA_TFLOPS, 6 * tf, 6 * tf / PEAK_BF16_MFMA_TFLOPS)]
    for fam, title in (('conv_igemm_fwd', 'direct convolution, forward (algorithmic FLOPs)'),
                       ('conv_igemm_dgrad', 'direct convolution, data gradient (algorithmic FLOPs)'),
                       ('conv_wino2_fwd', 'fused Winograd F(2x2,3x3) kernel, forward (algorithmic conv FLOPs; executed MFMA FLOPs = x 16/36)'),
                       ('conv_wino2_dgrad', 'fused Winograd F(2x2,3x3) kernel, data gradient (algorithmic conv FLOPs; executed = x 16/36)'),
                       ('conv_wgrad', 'weight gradient, whole call incl. re-layout / transforms / reduce (ALGORITHMIC FLOPs = direct count; the wide layers execute a quarter of them in Winograd form, so "of peak" can exceed 1 here -- the bench line prices the family on executed FLOPs)'),
                       ('conv_wino_fwd', 'Winograd layer calls, forward: three kernels (ALGORITHMIC conv FLOPs: 4x the executed GEMM FLOPs, "of peak" > 1 is expected)'),
                       ('conv_wino_dgrad', 'Winograd layer calls, data gradient: three kernels (ALGORITHMIC conv FLOPs: 4x the executed GEMM FLOPs, "of peak" > 1 is expected)')):
        rr = rows(fam)
        if not rr:
            continue
        L += ['', '## %s' % title, '', '| launch | /step | avg us | ms/step | GFLOP | TFLOP/s | of peak | GB/s (algorithmic bytes) |',
              '|---|---|---|---|---|---|---|---|']
        tms = tfl = 0.0
        for tag, g in rr:
            us = 1e3 * g['ms'] / g['n']
            tf = g['flops'] / (g['ms'] * 1e-3) / 1e12 if g['ms'] > 0 else 0.0
            gb = g['bytes'] / (g['ms'] * 1e-3) / 1e9 if g['ms'] > 0 else 0.0
            tms += g['ms']; tfl += g['flops']
            L.append('| %s | %.1f | %.1f | %.3f | %.2f | %.1f | %.2f | %.0f |' % (
                tag, g['n'] / float(psteps), us, g['ms'] / psteps, g['flops'] / g['n'] / 1e9, tf, tf / PEAK_F32_MFMA_TFLOPS, gb))
        L += ['', 'all launches: %.2f ms/step, %.1f TFLOP/s = %.3f of peak' % (
            tms / psteps, tfl / (tms * 1e-3) / 1e12, tfl / (tms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS)]
    rr = rows('wino_transform')
    if rr:
        L += ['', '## Winograd transform kernels (bytes each pass streams)', '', '| launch | /step | avg us | ms/step | MB | GB/s | of 8 TB/s |',
              '|---|---|---|---|---|---|---|']
        tms = tby = 0.0
        for tag, g in rr:
            us = 1e3 * g['ms'] / g['n']
            gb = g['bytes'] / (g['ms'] * 1e-3) / 1e9 if g['ms'] > 0 else 0.0
            tms += g['ms']; tby += g['bytes']
            L.append('| %s | %.1f | %.1f | %.3f | %.1f | %.0f | %.2f |' % (tag, g['n'] / float(psteps), us, g['ms'] / psteps,
                                                                     g['bytes'] / g['n'] / 1e6, gb, gb / 8000.0))
        L += ['', 'all launches: %.2f ms/step, %.1f GB/step, %.0f GB/s' % (tms / psteps, tby / psteps / 1e9, tby / (tms * 1e-3) / 1e9)]
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, 'w') as f:
        f.write('\n'.join(L) + '\n')


def effective_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (the GPU box advertises 256 logical CPUs but the container is quota-limited)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(args):
    """Oracle step (CPU port of the reference loop body, literal call order) on the host cores: batch 2, one warm-up,
    then >= 3 timed iterations (bounded to ~30 s); min / median / mean iteration time reported."""
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    from oracle import nets as onets, steps as osteps
    from fcd_gan_pytorch_amd.synthetic import synthetic_tiles
    import fcd_gan_pytorch_amd as fcd
    C, H, W = args.bands, args.size, args.size
    n = 2
    cores = effective_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        sdD = fcd.Module.Discriminator_SRGAN_simple(C).state_dict()
        sdS = fcd.Module.Segmentor(C, bilinear=True).state_dict()
        sdG = fcd.Module.Generator(C).state_dict()
        sdV = fcd.Loss.PerceptionLoss(1, True, allow_seeded=True).net.state_dict()
    nets = osteps.Nets(sdG, sdS, sdD, sdV).make_optimizers('rsss')
    x, y, region = synthetic_tiles(1234, n, C, H, W)
    osteps.rsss_adversarial_step(nets, x, y, region)            # warm-up (oneDNN primitive creation)
    times, t_all = [], time.perf_counter()
    while len(times) < 3 or (len(times) < 8 and (time.perf_counter() - t_all) < 20.0):
        t0 = time.perf_counter()
        osteps.rsss_adversarial_step(nets, x, y, region)
        times.append(time.perf_counter() - t0)
    dt = sum(times)
    ts = sorted(times)
    med = ts[len(ts) // 2] if len(ts) % 2 else 0.5 * (ts[len(ts) // 2 - 1] + ts[len(ts) // 2])
    model = ''
    try:
        with open('/proc/cpuinfo') as f:
            model = [l.split(':', 1)[1].strip() for l in f if l.startswith('model name')][0]
    except Exception:
        pass
    return {'value': n / med, 'unit': 'tile-pairs/s', 'cores': cores, 'kind': 'port',
            'sample': '%d timed iterations of the literal Demo_RSSS adversarial step, batch %d, %dx%dx%d, '
                      'torch CPU (oneDNN) fp32, %d threads, after 1 warm-up; value = batch / median iteration time'
                      % (len(times), n, H, W, C, cores),
            'value_best': n / ts[0], 'value_mean': n * len(times) / dt,
            'iteration_s': {'min': ts[0], 'median': med, 'max': ts[-1], 'n': len(times)},
            'cpu_model': model, 'seconds': dt}


def resolve_launch(gpus, env):
    """How this invocation relates to the rank set: ('single', 1, 0, 0) -- one process, one GPU;
    ('worker', world, rank, local_rank) -- one rank of a torchrun launch; ('spawn', gpus, 0, 0) -- `--gpus N > 1`
    outside a launcher: bench.py must start the N ranks itself.

    ``--gpus`` is authoritative.  A launcher environment is recognised by RANK + LOCAL_RANK + WORLD_SIZE together
    (what torch.distributed.run exports); a stale WORLD_SIZE alone (e.g. inherited from an outer job) is ignored, and a
    live launcher whose WORLD_SIZE disagrees with --gpus is an error rather than a mis-reported n_gpus."""
    have = all(k in env for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'))
    if have:
        world, rank, local = int(env['WORLD_SIZE']), int(env['RANK']), int(env['LOCAL_RANK'])
        if world != gpus:
            raise SystemExit('bench.py: --gpus %d but the launcher environment says WORLD_SIZE=%d (RANK=%d): '
                             'launch `torch.distributed.run --nproc-per-node %d bench.py --gpus %d`, or unset '
                             'RANK/LOCAL_RANK/WORLD_SIZE' % (gpus, world, rank, gpus, gpus))
        if world == 1:
            return ('single', 1, 0, 0)
        return ('worker', world, rank, local)
    if gpus > 1:
        return ('spawn', gpus, 0, 0)
    return ('single', 1, 0, 0)


def spawn_ranks(gpus, argv):
    """Re-exec this script as `gpus` ranks through torch.distributed.run (the launch line the driver uses for N > 1);
    rank 0's JSON line reaches our stdout unchanged.  Returns the launcher's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'GROUP_RANK', 'ROLE_RANK'):
        env.pop(k, None)                               # a stale partial launcher environment must not leak into the children
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC: the only mode the host driver supports (RCCL needs it)
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def latest_traffic_file():
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_hbm_traffic.json')))
    return files[-1] if files else None


def _flush_c_stdio():
    import ctypes
    try:
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='rsss', choices=sorted(WORKLOADS),
                    help="'rsss' = the headline (BASELINE.json configs[2]/[3]); 'usss_g' = configs[1] "
                         "(G-only, 16 x 256x256x4); 'wsss' = configs[4] (512x512x3)")
    ap.add_argument('--batch', type=int, default=None, help='tile pairs per GPU per step (default: per workload)')
    ap.add_argument('--bands', type=int, default=None)
    ap.add_argument('--size', type=int, default=None)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-prof', action='store_true', help='skip the second (profiled) pass')
    ap.add_argument('--no-alt', action='store_true', help='skip the extra pass with the Winograd GEMMs on the fp32 matrix pipe')
    ap.add_argument('--prof-steps', type=int, default=3, help='steps of the profiled pass (HIP events around every launch)')
    ap.add_argument('--layers-md', default=None, help='write per-layer tables of the profiled pass to this markdown file')
    ap.add_argument('--no-throttle', action='store_true', help='switch LAUNCH_WINDOW=0: let the host run ahead until the hardware queue is full (it then '
                    'spins in the launch calls: one busy core per rank); default: at most 384 launches ahead, sleeping while it waits')
    ap.add_argument('--force-exchange', action='store_true', help='with --gpus 1: create a ONE-rank process group and run every '
                    'data-parallel collective through it (dp.force_exchange) -- the RCCL rehearsal a 1-GPU box allows')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend: 'nccl' (= RCCL over xGMI); "
                    "'gloo' only for functional tests of the multi-rank path on a single-GPU box")
    args = ap.parse_args()
    _, d_bands, d_size, d_batch, wl_desc = WORKLOADS[args.workload]
    args.bands = args.bands or d_bands
    args.size = args.size or d_size
    args.batch = args.batch or d_batch

    mode, world, rank, local_rank = resolve_launch(args.gpus, os.environ)
    if mode == 'spawn':
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the product has no CPU fallback')
    ndev = torch.cuda.device_count()
    if world > 1 and args.backend == 'nccl' and ndev < world:
        raise SystemExit('bench.py: --gpus %d over RCCL needs %d GPUs, this node shows %d (for a functional test of the '
                         'multi-rank path on fewer GPUs use --backend gloo)' % (world, world, ndev))
    dev_index = local_rank if args.backend == 'nccl' else local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    forced = world == 1 and args.force_exchange
    if world > 1 or forced:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if forced:
            import socket
            s = socket.socket(); s.bind(('127.0.0.1', 0)); os.environ['MASTER_PORT'] = str(s.getsockname()[1]); s.close()
            os.environ['RANK'], os.environ['WORLD_SIZE'] = '0', '1'
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.backend)
    n_gpus = world

    from fcd_gan_pytorch_amd import _lib, dp as fdp
    if forced:
        fdp.force_exchange(True)
    if args.no_throttle:
        _lib.set_switch('LAUNCH_WINDOW', 0)
    step, opts = build_workload(args, dev, rank)

    def barrier():
        if world > 1 or forced:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # ---- headline: K steps, NO per-launch events (the profiler is a separate pass below)
    cpu0, thr0, t0 = time.process_time(), time.thread_time(), time.perf_counter()
    for _ in range(args.steps):
        out = step()
    # host time the step's launches cost: CPU seconds of the whole process (Python thread + autograd engine thread + RCCL
    # proxy threads) up to the point where everything is queued, i.e. before the blocking synchronize
    cpu_queued, thr_queued, t_queued = time.process_time() - cpu0, time.thread_time() - thr0, time.perf_counter() - t0
    barrier()
    dt = time.perf_counter() - t0
    cpu_all = time.process_time() - cpu0
    losses = {k: float(v.detach()) for k, v in out.items() if v.dim() == 0}
    host = {'host_cpu_ms_per_step': 1e3 * cpu_queued / args.steps,
            'host_cpu_ms_per_step_python_thread': 1e3 * thr_queued / args.steps,
            'host_wall_ms_per_step_until_queued': 1e3 * t_queued / args.steps,
            'host_cpu_ms_per_step_incl_final_sync': 1e3 * cpu_all / args.steps}
    exch = {k: o.last_exchange for k, o in opts.items()}
    if world > 1:
        t = torch.tensor([dt, host['host_cpu_ms_per_step']], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0].item())
        host['host_cpu_ms_per_step_max_over_ranks'] = float(t[1].item())
        # every rank's exchange record: all ranks must have cut the same buckets and left the same number during backward
        early = torch.tensor([(e or {}).get('launched_during_backward', -1) for e in exch.values()], dtype=torch.int64, device=dev)
        lo, hi = early.clone(), early.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        for i, k in enumerate(exch):
            if exch[k] is not None:
                exch[k] = dict(exch[k], launched_during_backward_min_over_ranks=int(lo[i]), launched_during_backward_max_over_ranks=int(hi[i]))
    # ---- the same K' steps with the Winograd GEMMs on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32) instead of the
    #      split-bf16 one: reported next to `value`, never as `value`
    alt = None
    pipes = {'wino_split': int(_lib.lib.fcd_conv_wino_split_set(-1) != 0), 'wgrad_split': int(_lib.lib.fcd_conv_wgrad_split_set(-1) != 0)}
    if not args.no_alt and (pipes['wino_split'] or pipes['wgrad_split']):
        _lib.lib.fcd_conv_wino_split_set(0)
        _lib.lib.fcd_conv_wgrad_split_set(0)
        ksteps = max(1, min(args.steps, 3))
        step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(ksteps):
            step()
        barrier()
        dta = time.perf_counter() - t1
        _lib.lib.fcd_conv_wino_split_set(pipes['wino_split'])
        _lib.lib.fcd_conv_wgrad_split_set(pipes['wgrad_split'])
        if world > 1:
            t = torch.tensor([dta], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dta = float(t.item())
        alt = {'what': 'same workload with fcd_conv_wino_split_set(0) AND fcd_conv_wgrad_split_set(0): every matrix instruction of the step on the '
                       'fp32 pipe (Winograd batched GEMMs on v_mfma_f32_32x32x2_f32, NCHW-direct 3x3 weight gradient on v_mfma_f32_32x32x2_f32; the '
                       'stride-2 weight gradients through channel-minor copies + the fp32 kernel)',
               'pipes': {'wino_split': 0, 'wgrad_split': 0},
               'value': args.batch * n_gpus * ksteps / dta, 'unit': 'tile-pairs/s', 'ms_per_step': 1e3 * dta / ksteps, 'steps': ksteps}
    # ---- profiled pass (not part of `value`): every launch bracketed by HIP events on its stream
    prof, detail, dt_prof, psteps = {}, [], None, 0
    if not args.no_prof:
        psteps = max(1, min(args.steps, args.prof_steps))
        step()                                                 # (re-pack after the fp32-pipe pass, outside the profiled region)
        _lib.prof_read(reset=True)
        _lib.lib.fcd_prof_enable(2 if (args.layers_md and rank == 0) else 1)
        barrier()
        t1 = time.perf_counter()
        for _ in range(psteps):
            step()
        barrier()
        dt_prof = time.perf_counter() - t1
        _lib.lib.fcd_prof_enable(0)
        prof = _lib.prof_read(reset=True)
        detail = _lib.prof_detail(reset=True)
        if args.layers_md and rank == 0:
            write_layer_tables(args.layers_md, detail, psteps, args)

    if rank == 0:
        total_pairs = args.batch * n_gpus * args.steps
        res = {
            'metric': 'tile-pairs/sec (train step) on 256x256x13 OSCD',
            'value': total_pairs / dt, 'unit': 'tile-pairs/s', 'n_gpus': n_gpus, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s, %d bands %dx%d, random-init weights, seeded (not ImageNet) VGG16 filters'
                                   % (wl_desc, args.bands, args.size, args.size),
                       'baseline_config': 'BASELINE.json configs[%d]' % WORKLOADS[args.workload][0],
                       'tile_pairs_per_gpu': args.batch, 'global_batch': args.batch * n_gpus,
                       'parallelism': 'dp%d' % n_gpus, 'bn': 'per-replica statistics',
                       'arithmetic': arithmetic_text(pipes['wino_split'], pipes['wgrad_split']), 'pipes': pipes,
                       'world_size': dist.get_world_size() if world > 1 else 1,
                       'backend': (dist.get_backend() if (world > 1 or forced) else 'none'),
                       'grad_exchange': ('bucketed all-reduce overlapped with backward' if world > 1 else
                                         'FORCED through a one-rank %s group (rehearsal of the collectives, --force-exchange)' % args.backend
                                         if forced else 'none (1 rank)'),
                       'grad_exchange_last_step': exch},
            'losses_last_step': losses,
            'peak_memory_bytes': {'allocated': int(torch.cuda.max_memory_allocated(dev)), 'reserved': int(torch.cuda.max_memory_reserved(dev)),
                                  'note': 'torch caching allocator, this rank, whole run (all passes); of 288 GB HBM3E'},
            'launch': 'launch by launch from Python (ctypes + autograd engine)',
            'host': dict(host, cores_usable=effective_cores(), launch_window=_lib.switch('LAUNCH_WINDOW'),
                         note='CPU seconds this rank spent issuing one step (all threads of the process) vs the step time: with N ranks '
                              'per node the sum over ranks has to fit the node\'s usable cores x ms_per_step'),
        }
        if alt:
            res['fp32_mfma_only'] = alt
        if prof:
            fwd, dg, wg = prof['conv_igemm_fwd'], prof['conv_igemm_dgrad'], prof['conv_wgrad']
            zero = dict(ms=0.0, launches=0, flops=0.0, bytes=0.0)
            wf, wd = prof.get('conv_wino_fwd', zero), prof.get('conv_wino_dgrad', zero)
            w2f, w2d = prof.get('conv_wino2_fwd', zero), prof.get('conv_wino2_dgrad', zero)
            wgemm, wxf = prof.get('wino_gemm', zero), prof.get('wino_transform', zero)
            wsplit = prof.get('wino_gemm_bf16x6', zero)

            cands = roofline_entries(prof, psteps, dt_prof)
            res['roofline'] = cands[0]
            res['roofline']['other_mfma_kernels'] = cands[1:]
            # HBM bytes per launch of the direct-conv family from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate
            # rocprofv3 passes over this very command: tools/pmc_bench.sh); bench.py cannot run the profiler on itself,
            # so it reports the committed measurement of the family it belongs to.
            tpath = latest_traffic_file()
            stamp = _lib.build_hash()            # the stamp baked into the LOADED binary, not a hash of whatever sources lie around
            if tpath and args.workload == 'rsss' and args.batch == 8 and args.bands == 13 and args.size == 256:
                with open(tpath) as f:
                    tjson = json.load(f)
                tj = tjson.get('families', {})
                fresh = tjson.get('kernel_source_hash') == stamp
                res['hbm_traffic_source'] = {
                    'file': os.path.relpath(tpath, ROOT), 'measured_on_kernel_source_hash': tjson.get('kernel_source_hash'),
                    'this_build_kernel_source_hash': stamp, 'valid_for_this_build': fresh,
                    'loaded_library': _lib.LIB_PATH, 'sources_next_to_it_hash': _lib.kernel_source_hash(),
                    'note': 'bench.py cannot run rocprofv3 --pmc on itself: `traffic` is the committed PMC measurement of this very '
                            'command (tools/pmc_hbm.sh), reported ONLY while the HIP sources it was taken on are the ones built here; '
                            'otherwise traffic is null'}
                # like for like: the report holds HBM bytes per STEP of every kernel that runs inside a family's
                # calls (tools/kernel_families.py: one map for both tools); both figures below are per CALL as this run counts them
                from tools.kernel_families import FAMILIES, calls_per_step
                scopes = {k: dict(launches=v['launches'], bytes=v['bytes']) for k, v in prof.items()}

                def alg_bytes_per_call(fam):
                    _, add, sub = FAMILIES[fam]
                    by = sum(scopes[s_]['bytes'] for s_ in add if s_ in scopes) - sum(scopes[s_]['bytes'] for s_ in sub if s_ in scopes)
                    n = calls_per_step(fam, scopes, 1)
                    return by / n if n > 0 else None
                for e in cands:
                    key = ('wino_gemm_split' if e['kernel'].startswith('wino_gemm_split') else
                           'wino_gemm' if e['kernel'].startswith('wino_gemm') else
                           'conv_wino2' if e['kernel'].startswith('conv_wino2') else
                           'conv_wgrad_split' if e['kernel'].startswith('conv_wgrad_roll_nchw') else
                           'conv_wgrad' if e['kernel'].startswith('weight gradient') else
                           'conv_igemm' if e['kernel'].startswith('conv_igemm') else None)
                    if key not in FAMILIES:
                        continue
                    if not key:
                        continue
                    alg_b = alg_bytes_per_call(key)
                    e['algorithmic_bytes_per_launch'] = alg_b
                    if fresh and key in tj and 'hbm_bytes_per_step' in tj[key]:
                        n_call = calls_per_step(key, scopes, psteps)
                        e['traffic'] = tj[key]['hbm_bytes_per_step'] / n_call if n_call > 0 else None
                        e['traffic_unit'] = ('HBM bytes per call (= per launch of this table): PMC FETCH_SIZE x %.2f + WRITE_SIZE x %.2f over every kernel '
                                             'of the family, per step, / %.1f calls per step (factors calibrated in the same session on known '
                                             '2 GiB streams in the kernel\'s access pattern, %s)'
                                             % (tj[key]['fetch_factor'] or 1.0, tj[key]['write_factor'] or 1.0, n_call, os.path.relpath(tpath, ROOT)))
                        e['traffic_over_algorithmic'] = e['traffic'] / alg_b if (alg_b and e['traffic']) else None
            if wf['launches'] + wd['launches'] > 0:
                wms, wfl = wf['ms'] + wd['ms'], wf['flops'] + wd['flops']
                from fcd_gan_pytorch_amd import _lib as _l
                mt = _l.lib.fcd_conv_wino_set(-1)
                res['winograd'] = {
                    'what': '3x3 / stride-1 layers with >= 128 GEMM rows and >= 64 reduction channels run as Winograd '
                            'F(%dx%d, 3x3): input transform + batched fp32 MFMA GEMM + output transform (three kernels per '
                            'layer call); results equal the direct kernels within fp32 transform rounding (<= 2e-5 relative, '
                            'tests/test_gpu_ops.py)' % (mt, mt),
                    'layer_calls_per_step': (wf['launches'] + wd['launches']) / psteps, 'ms_per_step': wms / psteps,
                    'algorithmic_conv_tflops': wfl / (wms * 1e-3) / 1e12,
                    'gemm_ms_per_step': (wgemm['ms'] + wsplit['ms']) / psteps, 'transform_ms_per_step': wxf['ms'] / psteps,
                    'transform_gbps': wxf['bytes'] / (wxf['ms'] * 1e-3) / 1e9 if wxf['ms'] > 0 else None,
                    'share_of_step_time': wms / (1e3 * dt_prof),
                }
                dms, dfl = fwd['ms'] + dg['ms'] + w2f['ms'] + w2d['ms'], fwd['flops'] + dg['flops'] + w2f['flops'] + w2d['flops']
                res['conv_fwd_dgrad_algorithmic_tflops'] = (dfl + wfl) / ((dms + wms) * 1e-3) / 1e12
            res['kernel_families'] = {
                k: {'ms_per_step': v['ms'] / psteps, 'launches_per_step': v['launches'] / psteps,
                    'tflops': (v['flops'] / (v['ms'] * 1e-3) / 1e12) if v['ms'] > 0 and v['flops'] > 0 else None,
                    'gbps': (v['bytes'] / (v['ms'] * 1e-3) / 1e9) if v['ms'] > 0 and v['bytes'] > 0 else None}
                for k, v in prof.items() if v['launches'] > 0}
            for k in ('wino_gemm', 'wino_gemm_bf16x6', 'wino_transform'):
                if k in res['kernel_families']:
                    res['kernel_families'][k]['nested_in'] = 'conv_wino_fwd + conv_wino_dgrad + conv_wgrad_wino'
            for k in ('conv_wgrad_wino', 'conv_wgrad_bf16x6'):
                if k in res['kernel_families']:
                    res['kernel_families'][k]['nested_in'] = 'conv_wgrad'
            # the whole step on both FLOP counts: algorithmic = direct-convolution FLOPs of every conv launch;
            # executed = what the MFMA units are really asked to do (Winograd layers: the batched GEMM, 1/4 of the direct
            # count + tile padding); both over the HEADLINE step time (events off)
            wgw = prof.get('conv_wgrad_wino', zero)
            alg = fwd['flops'] + dg['flops'] + wf['flops'] + wd['flops'] + wg['flops'] + w2f['flops'] + w2d['flops']
            exe = fwd['flops'] + dg['flops'] + wgemm['flops'] + wsplit['flops'] + (wg['flops'] - wgw['flops']) + (w2f['flops'] + w2d['flops']) * 16.0 / 36.0
            step_s = dt / args.steps
            res['whole_step'] = {'algorithmic_tflops': alg / psteps / step_s / 1e12,
                                 'executed_tflops': exe / psteps / step_s / 1e12,
                                 'frac_executed': exe / psteps / step_s / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                 'algorithmic_tflop_per_step': alg / psteps / 1e12,
                                 'profiled_ms_per_step': 1e3 * dt_prof / psteps,
                                 'note': 'value / ms_per_step are timed with the per-launch events OFF; the roofline and '
                                         'kernel_families numbers come from a second pass of %d step(s) with them on' % psteps}
        if n_gpus == 1 and not args.no_cpu_baseline and args.workload == 'rsss':
            res['cpu_baseline'] = cpu_baseline(args)
    # the JSON line is the LAST thing on stdout: RCCL / Gloo write their banners through C stdio, which -- redirected to a file or a
    # pipe -- is block-buffered and would otherwise come out at exit, behind the line.  Every rank flushes, then rank 0 prints.
    _flush_c_stdio()
    if world > 1 or forced:
        dist.barrier()
    if rank == 0:
        res['self_check'] = check_result_consistency(res)      # [] = every roofline entry priced against the pipe it runs on, text = switch state
        if res['self_check']:
            print('bench.py: inconsistent line: %s' % '; '.join(res['self_check']), file=sys.stderr)
        print(json.dumps(res), flush=True)
    if world > 1 or forced:
        dist.destroy_process_group()
        _flush_c_stdio()


if __name__ == '__main__':
    main()
