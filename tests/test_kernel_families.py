"""tools/kernel_families.py is the ONE map from bench.py's kernel families to HIP kernels (PMC bytes per
dispatch against algorithmic bytes per call gave a direct weight gradient at 0.25x its algorithmic traffic).  CPU checks: the
map names real kernels, leaves no convolution kernel out, parses names as the profilers print them, and counts calls the way
bench.py counts launches."""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import kernel_families as kf      # noqa: E402

CSRC = os.path.join(ROOT, 'fcd_gan_pytorch_amd', 'csrc')


def _kernels_by_file():
    out = {}
    for f in glob.glob(os.path.join(CSRC, '*.hip')):
        txt = open(f).read()
        names = re.findall(r'__global__[^;{]*?\bvoid\s+(\w+)\s*\(', txt, flags=re.S)
        out[os.path.basename(f)] = sorted(set(names))
    return out


def test_every_listed_kernel_exists_and_is_in_one_family():
    have = {k for ks in _kernels_by_file().values() for k in ks}
    seen = {}
    for fam, (names, add, sub) in kf.FAMILIES.items():
        assert add, fam
        for n in names:
            assert n in have, '%s: no __global__ %s in csrc' % (fam, n)
            assert n not in seen, '%s in %s and %s' % (n, seen[n], fam)
            seen[n] = fam


def test_no_convolution_kernel_is_left_out():
    """Every kernel of the convolution sources belongs to a family, except the filter packers (their own bench family) and the
    lab-only GEMM variant; a new kernel must be added to the map (or here) before the traffic figures mean anything."""
    not_in_a_traffic_family = {
        'pack_weights_kernel', 'pack_weights_s2_kernel', 'pack_weights_s2t_kernel', 'pack_weights_t_kernel', 'wino_filter_kernel', 'wino_filter_multi_kernel', 'wino2_pack_kernel',      # pack_weights
        'channel_sum_part_kernel', 'channel_sum_fin_kernel', 'channel_psum_fin_kernel',                                            # misc (bias gradients of frozen-filter calls)
    }
    for f, ks in _kernels_by_file().items():
        if not f.startswith('conv_'):
            continue
        for k in ks:
            assert kf.family_of(k) is not None or k in not_in_a_traffic_family, '%s (%s) is in no family' % (k, f)


def test_names_as_the_profilers_print_them():
    assert kf.family_of('void wino_gemm_split256_kernel<0>(WinoGemmArgs)') == 'wino_gemm_split'
    assert kf.family_of('void wino_gemm_split_kernel<2, 2, true>(WinoGemmArgs)') == 'wino_gemm_split'
    assert kf.family_of('wino_gemm_split_res_kernel(WinoGemmArgs)') == 'wino_gemm_split'
    assert kf.family_of('void wino_gemm_kernel<2, 2>(WinoGemmArgs)') == 'wino_gemm'
    assert kf.family_of('void (anonymous namespace)::conv_wino2_kernel<2, 0, 8, 2, 1>((anonymous namespace)::Wino2Args)') == 'conv_wino2'
    assert kf.family_of('void conv_igemm_rows16_kernel<9, 9, 1, 8, 8, 32>(ConvArgs)') == 'conv_igemm'
    assert kf.family_of('void small_fc_kernel<8>(SmallFcArgs)') == 'conv_igemm'
    assert kf.family_of('void conv3x3_dgrad_c1_mfma_kernel<true>(float const*, unsigned char const*)') == 'conv_igemm'
    assert kf.family_of('nchw_to_nhwc_v4_kernel(float const*, float const*, float*, int, int, int, float*)') == 'conv_wgrad'
    assert kf.family_of('wgrad_reduce_wide_kernel(float const*, float*, long long, int, long long, int)') == 'conv_wgrad'
    assert kf.family_of('void wino_oi_kernel<16, 1, 2, 256, 2>(WinoOiArgs)') == 'wino_transform'
    assert kf.family_of('wino_output_blk_kernel(WinoOutArgs)') == 'wino_transform'
    assert kf.family_of('bn_act_apply_kernel(float const*, float*)') is None
    # the aggregation key of tools/pmc_hbm.sh: namespace stripped, cut at the first '('
    assert kf.family_of('void conv_wino2_kernel<0, 1, 8, 2, 1>') == 'conv_wino2'


def test_calls_per_step_counts_what_bench_counts():
    prof = {'conv_wgrad': dict(launches=75), 'conv_wgrad_wino': dict(launches=45), 'conv_igemm_fwd': dict(launches=51),
            'conv_igemm_dgrad': dict(launches=39), 'wino_gemm_bf16x6': dict(launches=201), 'conv_wino2_fwd': dict(launches=39),
            'conv_wino2_dgrad': dict(launches=12)}
    assert kf.calls_per_step('conv_wgrad', prof, 3) == 10.0          # direct calls only: the Winograd-form ones are priced under the split GEMM
    assert kf.calls_per_step('conv_igemm', prof, 3) == 30.0
    assert kf.calls_per_step('wino_gemm_split', prof, 3) == 67.0
    assert kf.calls_per_step('conv_wino2', prof, 3) == 17.0
    # a family whose traffic is B bytes per step over n calls of a bytes each: ratio = B / (n a), whatever the dispatch count
    B, n, a = 3.7e9, kf.calls_per_step('conv_wgrad', prof, 3), 0.35e9
    assert abs(B / n / a - 1.057) < 1e-3
