#!/usr/bin/env python3
"""Turn the four aggregates of tools/pmc_hbm.sh into profiles/rNN_hbm_traffic.json: calibration factors (known bytes /
reported bytes per access pattern) and the corrected HBM bytes per launch of the step's kernel families."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from fcd_gan_pytorch_amd._lib import build_hash as kernel_source_hash      # noqa: E402  (stamp baked into the binary the counters were taken on)

d = sys.argv[1]
CAL_BYTES = float(2 << 30)


def load(name):
    with open(os.path.join(d, name, 'agg.json')) as f:
        return json.load(f)


cf, cw = load('pmc_calib_FETCH_SIZE'), load('pmc_calib_WRITE_SIZE')
bf, bw = load('pmc_bench_FETCH_SIZE'), load('pmc_bench_WRITE_SIZE')


def per_dispatch(agg, key, ctr):
    hit = [v for k, v in agg.items() if key in k]
    return (sum(v[ctr] for v in hit) / max(1, sum(v['dispatches'] for v in hit))) if hit else None


# rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB
unit = 1024.0
cal = {}
for pat in ('calib_copy_b128', 'calib_copy_b32', 'calib_read_b128', 'calib_copy_lds_dma'):
    f, w = per_dispatch(cf, pat, 'FETCH_SIZE'), per_dispatch(cw, pat, 'WRITE_SIZE')
    cal[pat] = {'reported_fetch_bytes': f * unit if f is not None else None, 'reported_write_bytes': w * unit if w is not None else None,
                'true_bytes_each_way': CAL_BYTES,
                'fetch_factor': CAL_BYTES / (f * unit) if f else None,
                'write_factor': (CAL_BYTES / (w * unit) if w and pat != 'calib_read_b128' else None)}
F_B128, F_B32, F_DMA = cal['calib_copy_b128']['fetch_factor'], cal['calib_copy_b32']['fetch_factor'], cal['calib_copy_lds_dma']['fetch_factor']
W_B128, W_B32 = cal['calib_copy_b128']['write_factor'], cal['calib_copy_b32']['write_factor']

from tools.kernel_families import FAMILIES, family_of      # noqa: E402  (one kernel -> family map for this report and bench.py)

# calibration pattern per family: (fetch factor, write factor, what the streams are)
PATTERN = {
    'wino_gemm': (F_DMA, W_B32, 'both operands by global_load_lds 16 B/lane; C tile by dword stores (128 B per 32 lanes)'),
    'wino_gemm_split': (F_DMA, W_B128, 'bf16 planes of U and fp32 V by global_load_lds 16 B/lane; C tile in 32 x 32 blocks by dwordx4 stores'),
    'wino_transform': (F_B128, W_B32, 'input side: float4 row reads, V rows as 128-B dword-store segments; output side / fused output -> input '
                       'kernel: 8- / 16-B reads of M, float4 row stores / V rows'),
    'conv_wino2': (F_B32, W_B32, 'fused F(2x2) kernel: dword patch loads (8 x 32 pixel blocks + halo), filter slabs by LDS-DMA (L2-resident), '
                   'dword / 8-B output stores'),
    'conv_wgrad': (F_DMA, W_B32, 'direct weight-gradient calls: re-layout passes, GEMM kernel (operands by LDS-DMA), split-K reduce / finish kernels'),
    'conv_wgrad_split': (F_DMA, W_B32, 'NCHW-direct 3x3 weight gradient on the bf16 pipe: x rows and dY slabs by LDS-DMA as they lie, partials by dword stores'),
    'conv_igemm': (F_B32, W_B32, 'direct forward / data-gradient calls: dword patch loads (+ small L2-resident filter slabs by LDS-DMA), dword stores'),
}
STEPS = int(os.environ.get('PMC_BENCH_STEPS', '3'))      # tools/pmc_hbm.sh: bench.py --steps 2 --warmup 1
out = {'source': 'tools/pmc_hbm.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over tools/hbm_calib.bin '
                 '(2 GiB known streams) and over `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof` (3 steps in total)',
       'kernel_source_hash': kernel_source_hash(),
       'counter_unit': 'KiB', 'calibration': cal, 'families': {}}
out['steps_profiled'] = STEPS
out['note'] = ('bytes are per STEP (all dispatches of the family\'s kernels / steps profiled); bench.py divides by ITS calls per step, so that '
               'traffic and algorithmic bytes are both per call (tools/kernel_families.py)')
for fam, (ff, wf, what) in PATTERN.items():
    fsum = sum(v['FETCH_SIZE'] for k, v in bf.items() if family_of(k) == fam) * unit
    wsum = sum(v['WRITE_SIZE'] for k, v in bw.items() if family_of(k) == fam) * unit
    nd = sum(v['dispatches'] for k, v in bf.items() if family_of(k) == fam)
    if not nd:
        continue
    out['families'][fam] = {'kernels': sorted(k for k in bf if family_of(k) == fam), 'dispatches_per_step': nd / float(STEPS), 'streams': what,
                            'fetch_factor': ff, 'write_factor': wf,
                            'hbm_bytes_per_step': (fsum * (ff or 1.0) + wsum * (wf or 1.0)) / STEPS,
                            'hbm_fetch_bytes_per_step': fsum * (ff or 1.0) / STEPS, 'hbm_write_bytes_per_step': wsum * (wf or 1.0) / STEPS}
print(json.dumps(out, indent=1))
