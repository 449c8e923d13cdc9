"""bench.py's description of itself: the roofline entries and ``config.arithmetic`` are generated from what ran --
the profile scopes and the library's pipe switches -- and every ``frac`` is quoted against the peak of the matrix pipe the kernel
really executes on.  Pure-function tests on synthetic profiles + a consistency check of every committed round-6 bench line."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _prof():
    f = lambda ms, n, fl: dict(ms=ms, launches=n, flops=fl, bytes=1.0)
    return {'conv_igemm_fwd': f(10.0, 30, 1.0e12), 'conv_igemm_dgrad': f(8.0, 20, 0.8e12),
            'conv_wgrad': f(30.0, 60, 3.0e12),              # whole family: nests the two below
            'conv_wgrad_wino': f(18.0, 20, 2.0e12), 'conv_wgrad_bf16x6': f(6.0, 25, 0.7e12),
            'wino_gemm': f(0.0, 0, 0.0), 'wino_gemm_bf16x6': f(90.0, 200, 15.0e12),
            'conv_wino2_fwd': f(30.0, 40, 4.5e12), 'conv_wino2_dgrad': f(28.0, 40, 4.5e12),
            'conv_wino_fwd': f(100.0, 100, 40e12), 'conv_wino_dgrad': f(80.0, 80, 30e12), 'wino_transform': f(50.0, 300, 0.0)}


def test_roofline_entries_price_each_family_on_its_own_pipe():
    import bench
    ents = bench.roofline_entries(_prof(), psteps=3, dt_prof=0.24)
    by = {e['kernel'].split(' ')[0]: e for e in ents}
    assert ents[0]['kernel'].startswith('wino_gemm_split')                       # sorted by share of the step
    assert 'wino_gemm_kernel' not in by                                          # no launches -> no entry
    split, wgs, wfp = by['wino_gemm_split256_kernel'], by['conv_wgrad_roll_nchw_kernel<split>'], by['weight']
    # bf16 families: SIX executed FLOPs per fp32-equivalent one, over the dense bf16 peak
    assert split['pipe'] == 'bf16x6' and split['peak'] == bench.PEAK_BF16_MFMA_TFLOPS
    assert split['achieved'] == pytest.approx(6 * 15.0e12 / 90e-3 / 1e12) and split['frac'] == pytest.approx(split['achieved'] / 2500.0)
    assert split['fp32_equivalent_tflops'] == pytest.approx(15.0e12 / 90e-3 / 1e12)
    assert wgs['pipe'] == 'bf16x6' and wgs['peak'] == 2500.0
    assert wgs['achieved'] == pytest.approx(6 * 0.7e12 / 6e-3 / 1e12) and wgs['launches_per_step'] == pytest.approx(25 / 3)
    # what is left of the weight-gradient family after its two nested parts is the fp32-pipe remainder, priced on 157.3
    assert wfp['pipe'] == 'fp32' and wfp['peak'] == bench.PEAK_F32_MFMA_TFLOPS
    assert wfp['achieved'] == pytest.approx((3.0 - 2.0 - 0.7) * 1e12 / 6e-3 / 1e12) and wfp['launches_per_step'] == pytest.approx(15 / 3)
    assert by['conv_wino2_kernel']['achieved'] == pytest.approx(9e12 * 16 / 36 / 58e-3 / 1e12)
    for e in ents:
        assert e['frac'] == pytest.approx(e['achieved'] / e['peak'])
    res = {'roofline': dict(ents[0], other_mfma_kernels=ents[1:])}
    assert bench.check_result_consistency(res) == []
    # a line that quotes a bf16-pipe kernel against the fp32 peak (what round 5's line did for the weight gradient) is caught
    bad = json.loads(json.dumps(res))
    bad['roofline']['other_mfma_kernels'][[e['kernel'] for e in ents[1:]].index(wgs['kernel'])].update(peak=157.3, pipe='fp32')
    assert any('disagree' in m or 'priced' in m for m in bench.check_result_consistency(bad))


def test_arithmetic_text_follows_the_switches():
    import bench
    both, none = bench.arithmetic_text(1, 1), bench.arithmetic_text(0, 0)
    assert 'v_mfma_f32_32x32x16_bf16' in both and 'NCHW-direct 3x3 weight-gradient' in both.split('v_mfma_f32_32x32x16_bf16')[1]
    assert 'bf16' not in none and 'fp32_mfma_only' not in none
    only_w = bench.arithmetic_text(1, 0)
    head, tail = only_w.split('v_mfma_f32_32x32x16_bf16')
    assert 'NCHW-direct' in head and 'NCHW-direct' not in tail and 'Winograd' in tail
    res = {'config': {'arithmetic': both, 'pipes': {'wino_split': 1, 'wgrad_split': 0}}}
    assert bench.check_result_consistency(res) == ['config.arithmetic does not describe config.pipes']


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r06_bench*.json'))) or [None])
def test_committed_round6_lines_are_self_consistent(path):
    if path is None:
        pytest.skip('no round-6 bench line committed yet')
    import bench
    with open(path) as f:
        txt = f.read().strip()
    res = json.loads(txt.splitlines()[-1]) if not txt.startswith('{\n') else json.loads(txt)
    assert bench.check_result_consistency(res) == [], path
    assert res.get('self_check') == []
    assert 'peak_memory_bytes' in res and res['peak_memory_bytes']['allocated'] > 0
    fo = res.get('fp32_mfma_only')
    if fo:
        assert fo['pipes'] == {'wino_split': 0, 'wgrad_split': 0} and fo['value'] < res['value']
