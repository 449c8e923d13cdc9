"""fcd_gan_pytorch_amd -- MI355X (gfx950) native hot path of FCD-GAN.

Drop-in surface (same names as the reference's top-level modules):
    from fcd_gan_pytorch_amd.Module import Generator, Segmentor, Discriminator_SRGAN_simple, ...
    from fcd_gan_pytorch_amd.Loss   import CNetLoss, CGeneratorLoss, PerceptionLoss, region_loss
    from fcd_gan_pytorch_amd.ssim   import MS_SSIM, SSIM, ms_ssim, ssim
or call ``install_as_reference_modules()`` once and keep the reference scripts'
``from Module import *`` / ``from Loss import *`` / ``from ssim import MS_SSIM`` lines.

Importing the package loads ``libfcdgan_hip.so`` (hand-written HIP kernels, C ABI in
include/fcdgan_hip.h) and raises if it is missing: there is no CPU/eager fallback.
"""
import sys

from . import _lib          # noqa: F401  (fails loudly when the HIP library is absent)
from . import Module, Loss, ssim, optim, steps, dp, tiles, metrics, datasets, demos  # noqa: F401

__version__ = '0.1.0'

from . import _ops as _ops_mod
_ops_mod.install_foreign_optimizer_hook()      # stock torch.optim steps (incl. fused=True ones, which bump no version counter) invalidate the filter packs


def install_as_reference_modules():
    """Register this package's modules under the reference's top-level names."""
    sys.modules['Module'] = Module
    sys.modules['Loss'] = Loss
    sys.modules['ssim'] = ssim


_MODULE_CACHES = ('_fcd_folded', '_fcd_folded_key', '_fcd_raw_filters', '_fcd_w1')


def invalidate_caches(*modules):
    """Drop every derived buffer (packed / Winograd-transformed filters, BatchNorm-folded filters, the summed first VGG
    filter) that ``modules`` and their parameters carry.  Never needed after ``optimizer.step()``, ``load_state_dict``, ``.to()``
    or any in-place op on the parameters -- those bump ``tensor._version``, which every cache is keyed on.  Needed only after
    an edit THROUGH ``param.data`` (``p.data.clamp_(-1, 1)``, the commented-out WGAN clip of the reference's Demo_RSSS.py:309-311):
    ``.data`` has its own version counter, so neither autograd nor this package can see the change."""
    from . import _ops
    for m in modules:
        _ops.invalidate_packs(list(m.parameters()) + list(m.buffers()))
        for sub in m.modules():
            for k in _MODULE_CACHES:
                sub.__dict__.pop(k, None)


def set_sync_batchnorm(enabled=True, group=None):
    """Optional SyncBN: train-mode BatchNorm statistics (and the two backward sums)
    are all-reduced over ``group`` so that a data-parallel run reproduces the single-device
    large-batch numerics of the reference.  Off by default (per-replica statistics, standard DDP
    semantics; the headline throughput is measured with it off)."""
    from . import _ops
    _ops.SYNC_BN['enabled'] = bool(enabled)
    _ops.SYNC_BN['group'] = group
