#!/usr/bin/env python3
"""Fixtures for fcd_gan_pytorch_amd/datasets.py from the REFERENCE's GDALDataset_RSS /
OSCD_Dataset_RSS / WHU_Dataset_WSS.order_reset (build container only; in-memory GDAL stand-in
from gen_golden_tiles.py, a temporary OSCD-style directory of empty files).  -> datasets.npz"""
import os
import random
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden_tiles as T  # noqa: E402


def main():
    T.install_stubs()
    sys.path.insert(0, T.REF)
    import data_utils as RD
    out = {}
    rng = np.random.default_rng(77)
    tmp = tempfile.mkdtemp()
    names = ['abudhabi', 'paris']
    sizes = {'abudhabi': (4, 70, 90), 'paris': (4, 55, 48)}
    open(os.path.join(tmp, 'train.txt'), 'w').write(','.join(names) + '\n')
    for nm in names:
        d = os.path.join(tmp, nm, 'ImagePair')
        os.makedirs(d)
        nb, ys, xs = sizes[nm]
        files = {nm + '_1': rng.integers(0, 3000, (nb, ys, xs)).astype(np.uint16),
                 nm + '_2': rng.integers(0, 3000, (nb, ys, xs)).astype(np.uint16),
                 nm + '-cm.tif': rng.integers(1, 3, (1, ys, xs)).astype(np.uint8),
                 nm + '-region.tif': (rng.integers(0, 2, (1, ys, xs)) * 255).astype(np.uint8)}
        for fn, arr in files.items():
            open(os.path.join(d, fn), 'w').close()
            T.SCENES[os.path.join(d, fn)] = arr
            out['%s/%s' % (nm, fn)] = arr
    ds = RD.OSCD_Dataset_RSS(tmp, 'train.txt', patch_size=(40, 32), overlap_padding=(4, 3))
    # os.listdir order decides which of the two images is X: record what the reference picked
    for i, nm in enumerate(names):
        out['%s/xname' % nm] = np.array(os.path.basename(ds.pathlist[i][0]))
    n = len(ds)
    out['len'] = np.array([n] + list(ds.cumlen), np.int64)
    out['eff'] = np.array([list(ds.EffRange(i)) for i in range(n)], np.int64)
    pick = sorted(set([0, 1, ds.cumlen[0] - 1, ds.cumlen[0], n - 1]))
    out['pick'] = np.array(pick, np.int64)
    for it in pick:
        x, y, item, ref, region = ds[it]
        out['item%d/x' % it] = x.numpy(); out['item%d/y' % it] = y.numpy()
        out['item%d/ref' % it] = ref.numpy(); out['item%d/region' % it] = region.numpy()
        out['item%d/item' % it] = np.array(int(item))
    # pairing orders
    for tag, (c, nc, seed) in {'p1': (7, 3, 5), 'p2': (4, 10, 6), 'p3': (5, 5, 7)}.items():
        w = object.__new__(RD.WHU_Dataset_WSS)
        w.cds_len, w.ncds_len = c, nc
        random.seed(seed)
        w.order_reset()
        out[tag + '/meta'] = np.array([c, nc, seed])
        out[tag + '/cds'] = np.array(w.cds_order); out[tag + '/ncds'] = np.array(w.ncds_order)
    np.savez_compressed(os.path.join(HERE, 'datasets.npz'), **out)
    print('wrote datasets.npz', len(out))


if __name__ == '__main__':
    main()
