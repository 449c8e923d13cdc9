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

FAM = {
    # family: (kernel-name substring(s), fetch factor, write factor, what the streams are)
    'wino_gemm': (['wino_gemm_kernel'], F_DMA, W_B32, 'both operands by global_load_lds 16 B/lane; C tile by dword stores (128 B per 32 lanes)'),
    'wino_gemm_split': (['wino_gemm_split'], F_DMA, W_B128, 'bf16 planes of U and fp32 V by global_load_lds 16 B/lane; C tile by dword stores'),
    'wino_input_transform': (['wino_input_kernel', 'wino_input_roll_kernel', 'wino_wg_input_kernel', 'wino_wg_dy_kernel'], F_B128, W_B32, 'float4 row reads; V rows as 128 B dword-store segments'),
    'wino_output_transform': (['wino_output_kernel', 'wino_output_blk_kernel'], F_B32, W_B128, 'coalesced dword reads of M; float4 row stores'),
    'conv_wino2': (['conv_wino2_kernel'], F_B32, W_B32, 'fused F(2x2) kernel: dword patch loads (8 x 32 pixel blocks + halo), filter slabs by LDS-DMA (L2-resident), '
                   'dword / 8-B output stores'),
    'conv_wgrad': (['conv_wgrad_kernel', 'conv_wgrad_roll_kernel', 'conv_wgrad_thin_kernel', 'nchw_to_nhwc', 'wgrad_reduce'], F_DMA, W_B32,
                   'direct weight-gradient kernels incl. their re-layout and split-K reduce passes'),
    'conv_igemm': (['conv_igemm_kernel', 'conv_igemm_glds_kernel', 'conv3x3_fwd_thin', 'conv3x3_dgrad_thin'], F_B32, W_B32, 'dword patch loads (+ small L2-resident filter slabs by LDS-DMA); dword stores'),
}
out = {'source': 'tools/pmc_hbm.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over tools/hbm_calib.bin '
                 '(2 GiB known streams) and over `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof` (3 steps in total)',
       'kernel_source_hash': kernel_source_hash(),
       'counter_unit': 'KiB', 'calibration': cal, 'families': {}}
for fam, (keys, ff, wf, what) in FAM.items():
    fsum = sum(v['FETCH_SIZE'] for k, v in bf.items() if any(s in k for s in keys)) * unit
    wsum = sum(v['WRITE_SIZE'] for k, v in bw.items() if any(s in k for s in keys)) * unit
    nd = sum(v['dispatches'] for k, v in bf.items() if any(s in k for s in keys))
    if not nd:
        continue
    out['families'][fam] = {'dispatches': nd, 'streams': what, 'raw_fetch_bytes_per_launch': fsum / nd, 'raw_write_bytes_per_launch': wsum / nd,
                            'fetch_factor': ff, 'write_factor': wf,
                            'hbm_bytes_per_launch': (fsum * (ff or 1.0) + wsum * (wf or 1.0)) / nd,
                            'hbm_fetch_bytes_per_launch': fsum * (ff or 1.0) / nd, 'hbm_write_bytes_per_launch': wsum * (wf or 1.0) / nd}
print(json.dumps(out, indent=1))
