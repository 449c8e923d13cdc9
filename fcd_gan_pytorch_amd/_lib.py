"""ctypes binding of ``libfcdgan_hip.so`` (C ABI declared in include/fcdgan_hip.h).

There is NO fallback: if the shared library is missing or a symbol cannot be
resolved, importing this module raises -- the product never silently routes
around the HIP kernels.
"""
import ctypes
import os
import subprocess

# The host framework must be in the process BEFORE libfcdgan_hip.so: the PyTorch-ROCm wheel bundles
# its own libamdhip64.so.7, and the C library has to bind to that same HIP runtime instance (the
# streams and device pointers it is handed belong to it).  Loaded first, the wheel's copy satisfies
# our DT_NEEDED by soname; loaded second, the process would end up with two HIP runtimes and every
# launch from this library fails with "no ROCm-capable device is detected".
import torch  # noqa: F401  (import order matters, see above)

from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('FCD_LIB') or os.path.join(_HERE, 'libfcdgan_hip.so')   # FCD_LIB: A/B kernel builds
CSRC = os.path.join(_HERE, 'csrc')


class FcdError(RuntimeError):
    pass


def build(force=False, verbose=False):
    """Compile the HIP sources in-tree with hipcc for gfx950 (works without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    cmd = ['make', '-C', CSRC, '-j', str(max(2, os.cpu_count() or 2))]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
    if r.returncode != 0 or not os.path.exists(LIB_PATH):
        raise FcdError('building libfcdgan_hip.so failed (make -C %s)' % CSRC)
    return LIB_PATH


def kernel_source_hash():
    """Stamp of the HIP sources + the C-ABI header lying next to the package (csrc/source_hash.py: the same function the
    Makefile bakes into the binary).  ``build_hash()`` is the stamp of the LOADED binary; the two differ when the .so is
    stale or was substituted through FCD_LIB."""
    import runpy
    return runpy.run_path(os.path.join(CSRC, 'source_hash.py'))['source_hash']()


class WinoFwdExtras(Structure):
    _fields_ = [('v_keep', c_void_p), ('bn_part', c_void_p), ('bn_groups', c_int32), ('in_groups', c_int32),
                ('in_scale', c_void_p), ('in_shift', c_void_p)]


class ConvDesc(Structure):
    _fields_ = [(n, c_int32) for n in ('N', 'C', 'H', 'W', 'K', 'R', 'S', 'stride', 'pad', 'P', 'Q')]


P = c_void_p  # device pointers travel as plain addresses

_SIGS = {
    'fcd_version': (c_int, []),
    'fcd_build_hash': (c_char_p, []),
    'fcd_last_error_string': (c_char_p, []),
    'fcd_switch_count': (c_int, []),
    'fcd_switch_name': (c_char_p, [c_int]),
    'fcd_switch_help': (c_char_p, [c_int]),
    'fcd_switch_default': (c_int, [c_int]),
    'fcd_switch_get': (c_int, [c_char_p]),
    'fcd_switch_set': (c_int, [c_char_p, c_int]),
    'fcd_conv_packed_elems': (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    'fcd_conv_pack_weights': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'fcd_conv2d_fwd': (c_int, [POINTER(ConvDesc), P, P, P, P, c_int, P]),
    'fcd_conv2d_fwd_ex': (c_int, [POINTER(ConvDesc), P, P, P, P, c_int, P, c_float, P, P]),
    'fcd_conv2d_relu_bits_bytes': (c_size_t, [POINTER(ConvDesc)]),
    'fcd_conv2d_fwd_relu_bits': (c_int, [POINTER(ConvDesc), P, P, P, P, P, P]),
    'fcd_conv2d_bwd_data_bits': (c_int, [POINTER(ConvDesc), P, P, P, P, P]),
    'fcd_conv1x1_head_plan': (c_int, [c_int, c_int, c_int, c_int]),
    'fcd_conv1x1_head_fwd': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    'fcd_conv1x1_head_bwd_ws_bytes': (c_size_t, [c_int, c_int]),
    'fcd_conv1x1_head_bwd': (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, P, c_size_t, P]),
    'fcd_conv1x1_head_bn_fwd': (c_int, [P, P, P, c_int, P, P, P, c_int, c_int, c_int, c_int, P]),
    'fcd_conv1x1_head_bn_bwd_ws_bytes': (c_size_t, [c_int, c_int, c_int]),
    'fcd_conv1x1_head_bn_bwd': (c_int, [P, P, P, P, P, P, P, P, c_int, P, P, P, P, P, c_int, c_int, c_int, P, c_size_t, P]),
    'fcd_conv_wino_plan': (c_int, [POINTER(ConvDesc), c_int]),
    'fcd_conv_wino_set': (c_int, [c_int]),
    'fcd_conv_wino_split_set': (c_int, [c_int]),
    'fcd_conv_wgrad_split_set': (c_int, [c_int]),
    'fcd_conv_wino_cat_ok': (c_int, [P]),
    'fcd_conv2d_fwd_wino_cat': (c_int, [P, P, P, c_int, P, P, P, c_int, P, c_size_t, P]),
    'fcd_conv2d_bwd_data_wino_cat': (c_int, [P, P, P, P, P, P, c_int, P, c_size_t, P]),
    'fcd_conv2d_bwd_weight_bias_cat': (c_int, [P, P, P, c_int, P, P, P, P, P, c_size_t, P]),
    'fcd_conv_wino_ws_bytes': (c_size_t, [POINTER(ConvDesc), c_int]),
    'fcd_conv_wino_keepv_bytes': (c_size_t, [POINTER(ConvDesc)]),
    'fcd_conv_wino_in_affine_ok': (c_int, [POINTER(ConvDesc)]),
    'fcd_conv_wino_bn_split': (c_int, [POINTER(ConvDesc), c_int]),
    'fcd_conv_wino_bn_part_bytes': (c_size_t, [POINTER(ConvDesc), c_int]),
    'fcd_conv2d_fwd_wino_x': (c_int, [POINTER(ConvDesc), P, P, P, P, c_int, P, P, P, c_size_t, P, P]),
    'fcd_conv2d_fwd_wino_cat_x': (c_int, [P, P, P, c_int, P, P, P, c_int, P, c_size_t, P, P]),
    'fcd_conv_wino_relu_bits_bytes': (c_size_t, [POINTER(ConvDesc)]),
    'fcd_conv2d_fwd_wino_relu_bits': (c_int, [POINTER(ConvDesc), P, P, P, P, P, P, c_size_t, P]),
    'fcd_conv2d_bwd_data_wino_bits': (c_int, [POINTER(ConvDesc), P, P, P, P, P, c_size_t, P]),
    'fcd_conv_wino_chain_ok': (c_int, [P, c_int, c_int]),
    'fcd_conv_wino_chain_ws_bytes': (c_size_t, [P, c_int, c_int]),
    'fcd_conv_wino_chain_bits_bytes': (c_size_t, [POINTER(ConvDesc)]),
    'fcd_conv2d_fwd_wino_chain': (c_int, [P, c_int, P, P, P, P, P, P, P, P, c_size_t, P]),
    'fcd_conv2d_bwd_data_wino_chain': (c_int, [P, c_int, P, P, P, P, P, P, P, c_size_t, P]),
    'fcd_conv2d_fwd_wino_keepv': (c_int, [POINTER(ConvDesc), P, P, P, P, c_int, P, P, P, c_size_t, P, P]),
    'fcd_conv2d_fwd_wino_cat_keepv': (c_int, [P, P, P, c_int, P, P, P, c_int, P, c_size_t, P, P]),
    'fcd_conv2d_bwd_weight_bias_v': (c_int, [POINTER(ConvDesc), P, P, P, P, P, P, c_size_t, P]),
    'fcd_conv_wino_filter_elems': (c_int64, [c_int, c_int, c_int, c_int]),
    'fcd_conv_wino_pack': (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    'fcd_conv_wino_pack_blocks': (c_int, [c_int, c_int, c_int]),
    'fcd_conv_wino_pack_multi': (c_int, [P, c_int, c_int, c_double, P]),
    'fcd_conv2d_fwd_wino': (c_int, [POINTER(ConvDesc), P, P, P, P, c_int, P, P, P, c_size_t, P]),
    'fcd_conv2d_bwd_data_wino': (c_int, [POINTER(ConvDesc), P, P, P, P, P, P, c_size_t, P]),
    'fcd_conv_s2_dgrad_plan': (c_int, [POINTER(ConvDesc)]),
    'fcd_conv_s2_dgrad_packed_elems': (c_int64, [c_int, c_int]),
    'fcd_conv_s2_dgrad_pack': (c_int, [P, P, c_int, c_int, P]),
    'fcd_conv2d_bwd_data_s2': (c_int, [POINTER(ConvDesc), P, P, P, P, P]),
    'fcd_conv_wino2_plan': (c_int, [POINTER(ConvDesc), c_int]),
    'fcd_conv_wino2_filter_elems': (c_int64, [c_int, c_int, c_int]),
    'fcd_conv_wino2_pack': (c_int, [P, P, c_int, c_int, c_int, P]),
    'fcd_conv2d_fwd_wino2': (c_int, [POINTER(ConvDesc), P, P, P, P, c_int, P, c_float, P, P, P, P]),
    'fcd_conv2d_bwd_data_wino2': (c_int, [POINTER(ConvDesc), P, P, P, P, P, P]),
    'fcd_conv2d_bwd_data': (c_int, [POINTER(ConvDesc), P, P, P, P, P]),
    'fcd_conv2d_fwd_relu_pool': (c_int, [POINTER(ConvDesc), P, P, P, P, P, P]),
    'fcd_conv2d_bwd_data_pooled': (c_int, [POINTER(ConvDesc), P, P, P, P, P]),
    'fcd_conv2d_bwd_weight_ws_bytes': (c_size_t, [POINTER(ConvDesc)]),
    'fcd_conv2d_bwd_weight': (c_int, [POINTER(ConvDesc), P, P, P, P, P, c_size_t, P]),
    'fcd_conv2d_bwd_weight_bias': (c_int, [POINTER(ConvDesc), P, P, P, P, P, P, c_size_t, P]),
    'fcd_channel_sum_ws_bytes': (c_size_t, [c_int]),
    'fcd_channel_sum': (c_int, [P, P, P, c_int, c_int, c_int, P, c_size_t, P]),
    'fcd_bn_act_ws_bytes': (c_size_t, [c_int, c_int]),
    'fcd_bn_act_fwd': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, P, c_float, c_float, c_int,
                               P, P, c_int, P, c_float, P, c_size_t, P]),
    'fcd_bn_act_fwd_replay': (c_int, [P, P, c_int, c_int, c_int, c_int, P, c_int, P, P, P, P, c_float, c_float, P, P, c_int, P,
                                      c_float, P, c_size_t, P]),
    'fcd_bn_act_bwd': (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, P, c_float, c_int, P, P,
                               c_int, P, c_float, P, P, P, P, c_size_t, P]),
    'fcd_bn_relu_pool_plan': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'fcd_bn_relu_pool_fwd': (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    'fcd_bn_relu_pool_bwd': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, c_size_t, P]),
    'fcd_bn_train_stats': (c_int, [P, c_int, c_int, c_int, c_int, P, c_int, P, P, P, P, c_float, c_float, P, P, P, P, P, c_size_t, P]),
    'fcd_bn_act_fwd_parts': (c_int, [P, P, c_int, c_int, c_int, c_int, P, c_int, P, P, P, P, c_float, c_float, P, P, c_int, P, c_float,
                                     P, c_size_t, P]),
    'fcd_bn_partial_stats': (c_int, [P, P, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    'fcd_bn_act_fwd_from_stats': (c_int, [P, P, c_int, c_int, c_int, c_int, P, c_double, P, P, P, P, c_float, c_float,
                                          P, P, c_int, P, c_float, P, c_size_t, P]),
    'fcd_bn_bwd_partial': (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, P, P, P, c_int, P, c_float, P, c_size_t, P]),
    'fcd_bn_bwd_from_sums': (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, c_double, P, P, P, P, c_int, P, c_float,
                                     P, c_size_t, P]),
    'fcd_maxpool2_fwd': (c_int, [P, P, c_int, c_int, c_int, P]),
    'fcd_maxpool2_bwd': (c_int, [P, P, P, c_int, c_int, c_int, P]),
    'fcd_maxpool2_bwd_add': (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    'fcd_upsample2x_fwd': (c_int, [P, P, c_int, c_int, c_int, P]),
    'fcd_upsample2x_bwd': (c_int, [P, P, c_int, c_int, c_int, P]),
    'fcd_avgpool2_pad_fwd': (c_int, [P, P, c_int, c_int, c_int, P]),
    'fcd_avgpool2_pad_bwd': (c_int, [P, P, c_int, c_int, c_int, P]),
    'fcd_pair_gap_diff_fwd': (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    'fcd_pair_gap_diff_bwd': (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    'fcd_masked_stack_fwd': (c_int, [P, P, P, P, c_int, P, P, c_int, c_int, c_int, P]),
    'fcd_masked_stack_bwd': (c_int, [P, P, P, P, P, c_int, P, P, P, P, P, P, c_int, c_int, c_int, P]),
    'fcd_normalize_tiles': (c_int, [P, P, P, P, P, c_int, c_int, c_int, P]),
    'fcd_masked_recon_ws_bytes': (c_size_t, [c_int]),
    'fcd_masked_recon_fwd': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    'fcd_masked_recon_bwd': (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'fcd_ratio_mean_fwd': (c_int, [P, c_int, c_float, c_int, P, P]),
    'fcd_ratio_mean_bwd': (c_int, [P, P, c_int, c_float, c_int, P, P, P]),
    'fcd_ssim_ws_bytes': (c_size_t, [c_int, c_int, c_int]),
    'fcd_ssim_level_fwd': (c_int, [P, P, P, c_int, P, c_int, c_int, c_int, c_float, c_float, P, c_size_t, P]),
    'fcd_ssim_level_bwd': (c_int, [P, P, P, c_int, P, P, P, P, c_int, c_int, c_int, c_float, c_float, P, c_size_t, P]),
    'fcd_adam_step': (c_int, [P, P, P, P, c_int64, c_float, c_float, c_float, c_float, c_float, c_int, c_float, P]),
    'fcd_rmsprop_step': (c_int, [P, P, P, c_int64, c_float, c_float, c_float, c_float, c_float, P]),
    'fcd_adam_step_h': (c_int, [P, P, P, P, c_int64, P, c_float, c_float, c_float, c_float, c_float, P]),
    'fcd_rmsprop_step_h': (c_int, [P, P, P, c_int64, P, c_float, c_float, c_float, c_float, P]),
    'fcd_prof_enable': (None, [c_int]),
    'fcd_prof_families': (c_int, []),
    'fcd_prof_detail_read': (c_int64, [c_char_p, c_int64, c_int]),
    'fcd_prof_read': (c_int, [POINTER(c_double), c_int]),
    'fcd_prof_family_name': (c_char_p, [c_int]),
}

EXPORTS = sorted(_SIGS)


def _load():
    if not os.path.exists(LIB_PATH):
        raise FcdError('libfcdgan_hip.so not found at %s -- run `python -c "import __graft_entry__ as g; '
                       'g.build()"` (needs hipcc); there is no CPU fallback.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise FcdError('libfcdgan_hip.so does not export %s (stale build?)' % name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def switch(name):
    """Current value of a run-time switch of csrc/switches.h (name with or without the FCD_ prefix)."""
    v = lib.fcd_switch_get(name.encode())
    if v < 0:
        raise FcdError('unknown switch %r' % name)
    return v


def set_switch(name, value):
    """Set a switch (``value`` < 0 or None: back to its default); returns the previous value."""
    old = lib.fcd_switch_set(name.encode(), -1 if value is None else int(value))
    if old < 0:
        raise FcdError('unknown switch %r' % name)
    return old


class switched:
    """``with switched(WGRAD_SPLIT=0, WINO_CHAIN=0): ...`` -- switches set for the block and restored after it (tests, A/B tools)."""

    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = set_switch(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_switch(k, v)
        return False


def switch_table():
    """[(name, value, default, help)] of every switch, in table order."""
    return [(lib.fcd_switch_name(i).decode(), lib.fcd_switch_get(lib.fcd_switch_name(i)), lib.fcd_switch_default(i),
             lib.fcd_switch_help(i).decode()) for i in range(lib.fcd_switch_count())]


def build_hash():
    """Source stamp baked into the loaded libfcdgan_hip.so at build time (``fcd_build_hash``)."""
    return lib.fcd_build_hash().decode()


def check(rc, what=''):
    if rc != 0:
        msg = lib.fcd_last_error_string()
        raise FcdError('%s failed (%d): %s' % (what or 'fcd call', rc, msg.decode() if msg else ''))


def prof_read(reset=True):
    """{family: dict(ms, launches, flops, bytes)} accumulated since the last reset."""
    nf = lib.fcd_prof_families()
    buf = (c_double * (nf * 4))()
    check(lib.fcd_prof_read(buf, 1 if reset else 0), 'fcd_prof_read')
    out = {}
    for f in range(nf):
        out[lib.fcd_prof_family_name(f).decode()] = dict(ms=buf[f * 4], launches=int(buf[f * 4 + 1]),
                                                         flops=buf[f * 4 + 2], bytes=buf[f * 4 + 3])
    return out


def prof_detail(reset=True):
    """Per-launch log of a profiled region (``lib.fcd_prof_enable(2)``; call :func:`prof_read` first):
    list of dict(family, tag, ms, flops, bytes)."""
    need = lib.fcd_prof_detail_read(None, 0, 0)
    buf = ctypes.create_string_buffer(int(need))
    lib.fcd_prof_detail_read(buf, need, 1 if reset else 0)
    out = []
    for line in buf.value.decode().splitlines():
        fam, tag, ms, fl, by = line.split('\t')
        out.append(dict(family=fam, tag=tag, ms=float(ms), flops=float(fl), bytes=float(by)))
    return out
