"""`make lab`: the lab kernels are the PRODUCT sources + csrc/lab/*.patch -- no forked copies that rot.  This
test is what notices a product edit the patches no longer apply to: it rebuilds the lab library from scratch and checks that it
exports the product's C ABI plus nothing the header does not declare."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'fcd_gan_pytorch_amd', 'csrc')


def _exports(path):
    out = subprocess.run(['nm', '-D', '--defined-only', path], stdout=subprocess.PIPE, text=True, check=True).stdout
    return sorted(l.split()[-1] for l in out.splitlines() if l.split()[-1].startswith('fcd_'))


def test_lab_is_product_plus_patches_and_still_builds():
    lab_dir = os.path.join(CSRC, 'lab')
    assert sorted(f for f in os.listdir(lab_dir) if f.endswith(('.hip', '.h'))) == [], 'a forked kernel source is back in csrc/lab'
    assert sorted(f for f in os.listdir(lab_dir) if f.endswith('.patch')) == ['conv_wino.patch', 'conv_wino2.patch']
    shutil.rmtree(os.path.join(CSRC, 'build', 'lab'), ignore_errors=True)
    lab = os.path.join(ROOT, 'fcd_gan_pytorch_amd', 'libfcdgan_hip_lab.so')
    if os.path.exists(lab):
        os.remove(lab)
    r = subprocess.run(['make', '-C', CSRC, 'lab', '-j', '8', 'LAB_FLAGS=-DFCD_SEXP=8 -DW2_EXP=2 -DFCD_YEXP=2'],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    try:
        # the patches touch kernels only: same C ABI as the product library
        assert _exports(lab) == _exports(os.path.join(ROOT, 'fcd_gan_pytorch_amd', 'libfcdgan_hip.so'))
        # and the patched copies really carry the lab switches
        src = open(os.path.join(CSRC, 'build', 'lab', 'conv_wino.hip')).read()
        assert 'FCD_SEXP' in src and 'wino_gemm_split_pp_kernel' in src
        assert 'FCD_SEXP' not in open(os.path.join(CSRC, 'conv_wino.hip')).read().replace('FCD_SEXP /', '')       # (a comment may name it)
        assert 'W2_EXP' in open(os.path.join(CSRC, 'build', 'lab', 'conv_wino2.hip')).read()
    finally:
        os.remove(lab)          # diagnostic builds give WRONG results: never leave one next to the product library
