#!/usr/bin/env python3
"""Headline benchmark: tile-pairs/s of the FCD-GAN RSSS adversarial train step
(Demo_RSSS.py:285-332: S fwd, D step, S step incl. eval-mode G, masked MSE,
MS-SSIM, per-band VGG16 perception, region losses, RMSprop) on 13-band 256x256
synthetic bi-temporal tiles, fp32, one process per GPU (weak scaling: fixed tiles
per GPU, gradients all-reduced over RCCL).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Prints ONE JSON line on rank 0 (see the driver contract) with two extra objects:
  roofline     -- the dominant kernel family (MFMA implicit-GEMM conv, forward +
                  data-gradient launches) timed with HIP events on the launch stream
                  over the timed region: algorithmic FLOPs / event time vs the fp32
                  MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md)
  cpu_baseline -- the CPU oracle (port of the reference's step on stock torch CPU ops)
                  timed on this box's host cores on a bounded sample (rank 0, N=1 only)
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


WORKLOADS = {
    # name: (BASELINE.json config, default bands, size, tile pairs per GPU, description)
    'rsss': (2, 13, 256, 8, 'Demo_RSSS adversarial step (S+D+G, masked MSE, MS-SSIM, per-band VGG16 perception, '
                            'region losses, RMSprop)'),
    'usss_g': (1, 4, 256, 16, 'Demo_USSS generator-only pre-training step (G fwd/bwd, masked L1 with cmap=0, MS-SSIM, '
                              'per-band VGG16 perception, Adam)'),
    'wsss': (4, 3, 512, 4, 'Demo_WSSS adversarial step (S on the changed and the unchanged pair, D, eval-mode G, masked '
                           'MSE, MS-SSIM, RGB VGG16 perception, RMSprop)'),
}


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
            crit = fcd.Loss.CNetLoss(channel=C, perception_layer=1, perception_perBand=True)      # Demo_USSS.py:116
        else:
            crit = fcd.Loss.CGeneratorLoss(channel=C, perception_layer=1, perception_perBand=args.workload == 'rsss')
    for m in (netD, netS, netG, crit):
        m.to(dev)
    multi = dist.is_initialized() and dist.get_world_size() > 1
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
        return step

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
        return step

    def step():
        return fcd.steps.rsss_adversarial_step(netS, netD, netG, crit, optS, optD, x, y, region)
    return step


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
    """Oracle step (CPU port of the reference loop body, literal call order) on the host cores."""
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    from oracle import nets as onets, steps as osteps
    from fcd_gan_pytorch_amd.synthetic import synthetic_tiles
    import fcd_gan_pytorch_amd as fcd
    C, H, W = args.bands, args.size, args.size
    n = 1
    cores = effective_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        sdD = fcd.Module.Discriminator_SRGAN_simple(C).state_dict()
        sdS = fcd.Module.Segmentor(C, bilinear=True).state_dict()
        sdG = fcd.Module.Generator(C).state_dict()
        sdV = fcd.Loss.PerceptionLoss(1, True).net.state_dict()
    nets = osteps.Nets(sdG, sdS, sdD, sdV).make_optimizers('rsss')
    x, y, region = synthetic_tiles(1234, n, C, H, W)
    osteps.rsss_adversarial_step(nets, x, y, region)            # warm-up (oneDNN primitive creation)
    iters, t0 = 0, time.perf_counter()
    while iters < 2 and (time.perf_counter() - t0) < 25.0 or iters == 0:
        osteps.rsss_adversarial_step(nets, x, y, region)
        iters += 1
    dt = time.perf_counter() - t0
    model = ''
    try:
        with open('/proc/cpuinfo') as f:
            model = [l.split(':', 1)[1].strip() for l in f if l.startswith('model name')][0]
    except Exception:
        pass
    return {'value': n * iters / dt, 'unit': 'tile-pairs/s', 'cores': cores, 'kind': 'port',
            'sample': '%d iteration(s) of the literal Demo_RSSS adversarial step, batch %d, %dx%dx%d, '
                      'torch CPU (oneDNN) fp32, %d threads, after 1 warm-up' % (iters, n, H, W, C, cores),
            'cpu_model': model, 'seconds': dt}


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
    ap.add_argument('--no-prof', action='store_true', help='do not bracket launches with HIP events')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend: 'nccl' (= RCCL over xGMI); "
                    "'gloo' only for functional tests of the multi-rank path on a single-GPU box")
    args = ap.parse_args()
    _, d_bands, d_size, d_batch, wl_desc = WORKLOADS[args.workload]
    args.bands = args.bands or d_bands
    args.size = args.size or d_size
    args.batch = args.batch or d_batch

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the product has no CPU fallback')
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.backend == 'nccl' else local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.backend)
    n_gpus = world

    from fcd_gan_pytorch_amd import _lib
    step = build_workload(args, dev, rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    if not args.no_prof:
        _lib.prof_read(reset=True)
        _lib.lib.fcd_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    _lib.lib.fcd_prof_enable(0)
    prof = _lib.prof_read(reset=True) if not args.no_prof else {}
    losses = {k: float(v) for k, v in out.items() if v.dim() == 0}
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        total_pairs = args.batch * n_gpus * args.steps
        res = {
            'metric': 'tile-pairs/sec (train step) on 256x256x13 OSCD',
            'value': total_pairs / dt, 'unit': 'tile-pairs/s', 'n_gpus': n_gpus, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s, %d bands %dx%d, random-init weights'
                                   % (wl_desc, args.bands, args.size, args.size),
                       'baseline_config': 'BASELINE.json configs[%d]' % WORKLOADS[args.workload][0],
                       'tile_pairs_per_gpu': args.batch, 'global_batch': args.batch * n_gpus,
                       'parallelism': 'dp%d' % n_gpus, 'bn': 'per-replica statistics'},
            'losses_last_step': losses,
        }
        if prof:
            fwd, dg, wg = prof['conv_igemm_fwd'], prof['conv_igemm_dgrad'], prof['conv_wgrad']
            zero = dict(ms=0.0, launches=0, flops=0.0, bytes=0.0)
            wf, wd = prof.get('conv_wino_fwd', zero), prof.get('conv_wino_dgrad', zero)
            wgemm, wxf = prof.get('wino_gemm', zero), prof.get('wino_transform', zero)

            def mfma_entry(name, ms, flops, launches, what):
                ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
                return {'bound': 'mfma', 'kernel': name, 'flops_counted': what, 'achieved': ach,
                        'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / PEAK_F32_MFMA_TFLOPS,
                        'traffic': None, 'launches_per_step': launches / args.steps,
                        'avg_launch_ms': ms / max(launches, 1), 'gflop_per_launch': flops / max(launches, 1) / 1e9,
                        'share_of_step_time': ms / (1e3 * dt) if dt > 0 else None}
            # the MFMA-bound kernels of the step, each priced on the FLOPs its launches really issue:
            #  * conv_igemm*: direct implicit-GEMM convolution -> algorithmic FLOPs 2 N K P Q C R S
            #  * wino_gemm: the batched GEMM of the Winograd path -> 2 (m+2)^2 rows Kc T (= the algorithmic conv
            #    FLOPs of those layers / 4 for F(4x4,3x3), plus tile padding)
            #  * conv_wgrad: weight gradient (re-layout + reduce passes included in its time)
            cands = [
                mfma_entry('conv_igemm_kernel / conv_igemm_glds_kernel (direct fp32 MFMA implicit-GEMM convolution; forward + '
                           'data-gradient launches)', fwd['ms'] + dg['ms'], fwd['flops'] + dg['flops'],
                           fwd['launches'] + dg['launches'], 'algorithmic convolution FLOPs'),
                mfma_entry('wino_gemm_kernel (batched fp32 MFMA GEMM of the Winograd F(4x4,3x3) path; forward, data-gradient '
                           'and weight-gradient launches)', wgemm['ms'], wgemm['flops'], wgemm['launches'], 'executed GEMM FLOPs'),
                mfma_entry('weight gradient: conv_wgrad_roll_kernel / conv_wgrad_kernel (+ re-layout, split-K reduce) and, for the wide '
                           '3x3 layers, the Winograd F(4x4,3x3) form (its GEMM launches are also part of wino_gemm_kernel above)',
                           wg['ms'], wg['flops'], wg['launches'], 'algorithmic weight-gradient FLOPs (direct count)'),
            ]
            cands.sort(key=lambda e: -(e['share_of_step_time'] or 0.0))
            res['roofline'] = cands[0]
            res['roofline']['other_mfma_kernels'] = cands[1:]
            # HBM bytes per launch of the direct-conv family from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate
            # rocprofv3 passes over this very command: tools/pmc_bench.sh); bench.py cannot run the profiler on itself,
            # so it reports the committed measurement of the family it belongs to.
            tpath = os.path.join(ROOT, 'profiles', 'r01_hbm_traffic.json')
            if os.path.exists(tpath) and args.workload == 'rsss' and args.batch == 8 and args.bands == 13 and args.size == 256:
                with open(tpath) as f:
                    tj = json.load(f)
                for e in cands:
                    key = 'wino_gemm' if e['kernel'].startswith('wino_gemm') else ('conv_igemm' if e['kernel'].startswith('conv_igemm') else None)
                    if key and key in tj:
                        e['traffic'] = tj[key]['hbm_bytes_per_launch']
                        e['traffic_unit'] = 'HBM bytes per launch (PMC, profiles/r01_hbm_traffic.json)'
                        e['algorithmic_bytes_per_launch'] = tj[key].get('algorithmic_bytes_per_launch')
            if wf['launches'] + wd['launches'] > 0:
                wms, wfl = wf['ms'] + wd['ms'], wf['flops'] + wd['flops']
                from fcd_gan_pytorch_amd import _lib as _l
                mt = _l.lib.fcd_conv_wino_set(-1)
                res['winograd'] = {
                    'what': '3x3 / stride-1 layers with >= 128 GEMM rows and >= 64 reduction channels run as Winograd '
                            'F(%dx%d, 3x3): input transform + batched fp32 MFMA GEMM + output transform (three kernels per '
                            'layer call); results equal the direct kernels within fp32 transform rounding (<= 2e-5 relative, '
                            'tests/test_gpu_ops.py)' % (mt, mt),
                    'layer_calls_per_step': (wf['launches'] + wd['launches']) / args.steps, 'ms_per_step': wms / args.steps,
                    'algorithmic_conv_tflops': wfl / (wms * 1e-3) / 1e12,
                    'gemm_ms_per_step': wgemm['ms'] / args.steps, 'transform_ms_per_step': wxf['ms'] / args.steps,
                    'transform_gbps': wxf['bytes'] / (wxf['ms'] * 1e-3) / 1e9 if wxf['ms'] > 0 else None,
                    'share_of_step_time': wms / (1e3 * dt),
                }
                dms, dfl = fwd['ms'] + dg['ms'], fwd['flops'] + dg['flops']
                res['conv_fwd_dgrad_algorithmic_tflops'] = (dfl + wfl) / ((dms + wms) * 1e-3) / 1e12
            res['kernel_families'] = {
                k: {'ms_per_step': v['ms'] / args.steps, 'launches_per_step': v['launches'] / args.steps,
                    'tflops': (v['flops'] / (v['ms'] * 1e-3) / 1e12) if v['ms'] > 0 and v['flops'] > 0 else None,
                    'gbps': (v['bytes'] / (v['ms'] * 1e-3) / 1e9) if v['ms'] > 0 and v['bytes'] > 0 else None}
                for k, v in prof.items() if v['launches'] > 0}
            for k in ('wino_gemm', 'wino_transform'):
                if k in res['kernel_families']:
                    res['kernel_families'][k]['nested_in'] = 'conv_wino_fwd + conv_wino_dgrad'
        if n_gpus == 1 and not args.no_cpu_baseline and args.workload == 'rsss':
            res['cpu_baseline'] = cpu_baseline(args)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
