"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads here
(no GPU needed), exports every symbol include/fcdgan_hip.h declares, and its
argument validation works without launching anything."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, 'include', 'fcdgan_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(fcd_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    from fcd_gan_pytorch_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), 'libfcdgan_hip.so does not export %s' % s
    # the ctypes table binds exactly the declared API
    assert sorted(_lib.EXPORTS) == syms


def test_argument_validation_without_gpu():
    from fcd_gan_pytorch_amd import _lib
    lib = _lib.lib
    assert lib.fcd_version() >= 100
    d = _lib.ConvDesc(1, 4, 8, 8, 8, 3, 3, 1, 1, 7, 7)      # wrong P,Q
    rc = lib.fcd_conv2d_fwd(ctypes.byref(d), 16, 16, None, 16, 0, None)
    assert rc == -1 and b'output size' in lib.fcd_last_error_string()
    d = _lib.ConvDesc(1, 4, 8, 8, 8, 5, 5, 1, 2, 8, 8)      # unsupported 5x5
    assert lib.fcd_conv2d_bwd_weight_ws_bytes(ctypes.byref(d)) == 0
    # 64 GEMM rows, 3x3: T layout [chunk of 4 channels][half][row padded to 128][20]
    assert lib.fcd_conv_packed_elems(64, 13, 3, 3, 0) == (16 // 4) * 2 * 128 * 20
    # data gradient of the same filter: 13 GEMM rows <= 32 -> row layout [(k, r, s)][rows padded to 128]
    assert lib.fcd_conv_packed_elems(64, 13, 3, 3, 1) == 64 * 9 * 128
    assert lib.fcd_conv_packed_elems(64, 13, 9, 9, 0) == 16 * 81 * 128
    assert lib.fcd_bn_act_ws_bytes(64, 2) > 0
    # Winograd layer plan (pure host logic): wide 3x3 / stride-1 layers only
    prev = lib.fcd_conv_wino_set(4)
    try:
        wide = _lib.ConvDesc(2, 256, 32, 32, 512, 3, 3, 1, 1, 32, 32)
        assert lib.fcd_conv_wino_plan(ctypes.byref(wide), 0) == 4 and lib.fcd_conv_wino_plan(ctypes.byref(wide), 1) == 4
        T = 2 * 8 * 8
        assert lib.fcd_conv_wino_ws_bytes(ctypes.byref(wide), 0) == 36 * (256 // 32) * T * 32 * 4 + 36 * 512 * T * 4 + 256
        assert lib.fcd_conv_wino_filter_elems(512, 256, 0, 4) == 36 * 512 * 256 * 5 // 2     # fp32 U + three bf16 planes
        assert lib.fcd_conv_wino_filter_elems(512, 256, 1, 2) == 16 * 256 * 512 * 5 // 2
        for d in (_lib.ConvDesc(2, 64, 32, 32, 64, 3, 3, 1, 1, 32, 32),        # 64 GEMM rows
                  _lib.ConvDesc(2, 256, 32, 32, 512, 3, 3, 2, 1, 16, 16),      # stride 2
                  _lib.ConvDesc(2, 256, 32, 32, 512, 1, 1, 1, 0, 32, 32),      # 1x1
                  _lib.ConvDesc(2, 13, 32, 32, 256, 3, 3, 1, 1, 32, 32)):      # 13 reduction channels
            assert lib.fcd_conv_wino_plan(ctypes.byref(d), 0) == 0
            assert lib.fcd_conv_wino_ws_bytes(ctypes.byref(d), 0) == 0
        assert lib.fcd_conv_wino_set(0) == 4 and lib.fcd_conv_wino_plan(ctypes.byref(wide), 0) == 0
        assert lib.fcd_conv_wino_set(2) == 0 and lib.fcd_conv_wino_plan(ctypes.byref(wide), 0) == 2
    finally:
        lib.fcd_conv_wino_set(prev)


def test_product_rejects_cpu_tensors():
    import pytest
    import torch
    from fcd_gan_pytorch_amd import Module, _lib
    g = Module.Generator(4)
    with pytest.raises(_lib.FcdError):
        g(torch.zeros(1, 4, 16, 16))


def test_binary_carries_the_stamp_of_the_sources_it_was_built_from():
    """the traffic stamp must identify the LOADED binary.  ``fcd_build_hash()`` is baked in at build time
    (csrc/Makefile -> build/build_hash.h); for the in-tree build it equals the hash of the sources lying next to it, so a
    stale .so shows up here, and bench.py compares a committed PMC measurement against the binary's stamp only."""
    import re as _re
    from fcd_gan_pytorch_amd import _lib
    got = _lib.build_hash()
    assert _re.fullmatch(r'[0-9a-f]{16}', got)
    if not os.environ.get('FCD_LIB'):
        assert got == _lib.kernel_source_hash(), 'libfcdgan_hip.so is stale: rebuild (make -C fcd_gan_pytorch_amd/csrc)'
