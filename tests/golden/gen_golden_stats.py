#!/usr/bin/env python3
"""Golden fixture for the dataset statistics: runs the REFERENCE's own
``CommonFunc.Dataset_mean`` / ``Dataset_std`` / ``Dataset_meanstd`` over the reference's ``GDALDataset``
(GDAL replaced by the in-memory stand-in of gen_golden_tiles.py).  Scenes contain all-zero pixels (nodata)
so that the valid-pixel mask of CommonFunc.py:446 matters.  Output: tests/golden/stats.npz."""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden_tiles as G            # noqa: E402  (stubs + fake GDAL)


def main():
    G.install_stubs()
    sys.path.insert(0, G.REF)
    import data_utils as RD
    import CommonFunc as RC
    out = {}
    rng = np.random.default_rng(77)
    for tag, nb, ys, xs, patch, pad in [('s1', 4, 90, 70, (40, 40), (5, 5)), ('s2', 13, 64, 64, (32, 32), (4, 4))]:
        x = rng.integers(1, 4000, (nb, ys, xs)).astype(np.uint16)
        y = rng.integers(1, 4000, (nb, ys, xs)).astype(np.uint16)
        x[:, :7, :] = 0; x[:, 30:41, 20:33] = 0          # nodata strips: excluded from both scenes' statistics
        G.SCENES['x'], G.SCENES['y'] = x, y
        ds = RD.GDALDataset('x', 'y', patch_size=patch, overlap_padding=pad)
        mx, my = RC.Dataset_mean(ds)
        sx, sy = RC.Dataset_std(ds, mx, my)
        with tempfile.TemporaryDirectory() as td:
            t1, t2 = os.path.join(td, 'a.txt'), os.path.join(td, 'b.txt')
            first = RC.Dataset_meanstd(t1, t2, ds)      # computes + writes
            again = RC.Dataset_meanstd(t1, t2, ds)      # reads the cache back
            out[tag + '/txt1'] = np.frombuffer(open(t1, 'rb').read(), dtype=np.uint8)
            out[tag + '/txt2'] = np.frombuffer(open(t2, 'rb').read(), dtype=np.uint8)
        out[tag + '/meta'] = np.array([nb, ys, xs, patch[0], patch[1], pad[0], pad[1]], np.int64)
        out[tag + '/x'], out[tag + '/y'] = x, y
        out[tag + '/mean_std'] = np.stack([mx.numpy(), sx.numpy(), my.numpy(), sy.numpy()]).astype(np.float64)
        out[tag + '/first'] = np.array(first, np.float64)
        out[tag + '/again'] = np.array(again, np.float64)
    np.savez_compressed(os.path.join(HERE, 'stats.npz'), **out)
    print('wrote stats.npz', len(out), 'arrays')


if __name__ == '__main__':
    main()
