"""ONE map from bench.py's MFMA / transform families to the HIP kernels that run inside their calls.

bench.py prices a family per CALL (one ``FcdProfScope`` of the C ABI = one launch in its tables: a weight-gradient call is its
re-layout passes + the GEMM kernel + the split-K reduce); the PMC report (tools/pmc_hbm_report.py) sees DISPATCHES.  Round 4
averaged PMC bytes over dispatches and algorithmic bytes over calls, and left kernels out of two families: direct weight
gradient 0.25x algorithmic.  Both tools now take the kernel list from here, the report stores bytes per STEP, and bench.py
divides by ITS number of calls per step -- like for like (tests/test_kernel_families.py).
"""

# family -> (kernel-name substrings, bench.py profile scopes whose launches are the family's calls [added], [subtracted])
FAMILIES = {
    'wino_gemm_split': (['wino_gemm_split_kernel', 'wino_gemm_split256_kernel', 'wino_gemm_split_res_kernel'],
                        ['wino_gemm_bf16x6'], []),
    'wino_gemm': (['wino_gemm_kernel'], ['wino_gemm'], []),
    'conv_wino2': (['conv_wino2_kernel'], ['conv_wino2_fwd', 'conv_wino2_dgrad'], []),
    # direct weight gradients incl. the 1x1 head: every kernel a fcd_conv2d_bwd_weight* / fcd_conv1x1_head_bwd call launches
    # [r6] the NCHW-direct 3x3 weight-gradient kernel on the bf16 matrix pipe (default; with the switch WGRAD_SPLIT=0 the same kernel
    # name runs its fp32 instantiation and its calls open no conv_wgrad_bf16x6 scope -- an A/B configuration this map does not serve)
    'conv_wgrad_split': (['conv_wgrad_roll_nchw_kernel'], ['conv_wgrad_bf16x6'], []),
    'conv_wgrad': (['conv_wgrad_kernel', 'conv_wgrad_roll_kernel', 'conv_wgrad_thin_kernel', 'conv_wgrad_thin_finish_kernel',
                    'conv_wgrad_thin9_kernel', 'conv_wgrad_thin9_finish_kernel', 'thin9_bias_part_kernel', 'thin9_bias_fin_kernel',
                    'nchw_to_nhwc_kernel', 'nchw_to_nhwc_v4_kernel', 'wgrad_reduce_kernel', 'wgrad_reduce_wide_kernel',
                    'head_wgrad_kernel', 'head_wgrad_final_kernel', 'head_bn_reduce_kernel', 'head_bn_final_kernel'],
                   ['conv_wgrad'], ['conv_wgrad_wino', 'conv_wgrad_bf16x6']),
    # what stays on the direct forward / data-gradient entry points
    'conv_igemm': (['conv_igemm_kernel', 'conv_igemm_glds_kernel', 'conv_igemm_rows16_kernel', 'conv_s2sub_glds_kernel', 'small_fc_kernel', 'small_fc_narrow_kernel',
                    'conv3x3_fwd_thin_kernel', 'conv3x3_dgrad_thin_kernel', 'conv3x3_dgrad_thin_v4_kernel', 'conv3x3_dgrad_c1_mfma_kernel',
                    'head_fwd_kernel', 'head_dgrad_kernel', 'head_fwd_bn_kernel', 'head_bn_apply_kernel'],
                   ['conv_igemm_fwd', 'conv_igemm_dgrad'], []),
    'wino_transform': (['wino_input_kernel', 'wino_input_roll_kernel', 'wino_output_kernel', 'wino_output_blk_kernel',
                        'wino_output_blk_bn_kernel', 'wino_oi_kernel', 'wino_wg_dy_kernel', 'wino_wg_input_kernel', 'wino_wg_final_kernel',
                        'wino_wg_splitsum_kernel', 'wino_psum_fin_kernel'],
                       ['wino_transform'], []),
}


def base_name(kernel):
    """Kernel name as the traces print it -> bare function name (no template arguments, namespace, parameter list)."""
    k = kernel.replace('(anonymous namespace)::', '')
    if k.startswith('void '):
        k = k[5:]
    for stop in ('<', '('):
        i = k.find(stop)
        if i >= 0:
            k = k[:i]
    return k.strip()


def family_of(kernel):
    b = base_name(kernel)
    hits = [f for f, (names, _, _) in FAMILIES.items() if b in names]
    if len(hits) > 1:
        raise ValueError('%s is in more than one family: %s' % (b, hits))
    return hits[0] if hits else None


def calls_per_step(family, prof, psteps):
    """Calls of ``family`` per step from bench.py's per-scope profile ({scope: dict(launches=...)})."""
    _, add, sub = FAMILIES[family]
    n = sum(prof[s]['launches'] for s in add if s in prof) - sum(prof[s]['launches'] for s in sub if s in prof)
    return n / float(psteps)
