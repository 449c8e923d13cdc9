#!/usr/bin/env python3
"""Golden fixtures for the data-side rows: runs the REFERENCE's own
``data_utils.GDALDataset`` / ``metrics.Evaluator`` / ``CommonFunc.write_changemap_gdal`` /
``CommonFunc.NORMALIZE`` in the build container.  GDAL is absent, so ``osgeo.gdal`` is
replaced by an in-memory stand-in exposing exactly the calls the reference makes
(Open / RasterXSize / RasterYSize / RasterCount / GetRasterBand(b).ReadAsArray(x,y,w,h) /
GetDriver().Create / WriteArray) over NumPy arrays -- the tiling arithmetic under test is the
reference's own code.  Output: tests/golden/tiles.npz."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = '/root/reference'

SCENES = {}      # path -> (bands, H, W) array


class _Band:
    def __init__(self, arr):
        self.arr = arr

    def ReadAsArray(self, x, y, w, h):
        return self.arr[y:y + h, x:x + w].copy()

    def WriteArray(self, a, x, y):
        self.arr[y:y + a.shape[0], x:x + a.shape[1]] = a


class _DS:
    def __init__(self, arr):
        self.arr = arr
        self.RasterCount, self.RasterYSize, self.RasterXSize = arr.shape

    def GetRasterBand(self, b):
        return _Band(self.arr[b - 1])

    def GetDriver(self):
        return _Driver()

    def GetGeoTransform(self):
        return (0, 1, 0, 0, 0, 1)

    def GetProjection(self):
        return ''

    def SetGeoTransform(self, *a):
        pass

    def SetProjection(self, *a):
        pass


class _Driver:
    def Create(self, path, xs, ys, nb, dtype):
        SCENES[path] = np.zeros((nb, ys, xs), np.float32)
        return _DS(SCENES[path])


def install_stubs():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m
    tv = mod('torchvision'); tv.transforms = mod('torchvision.transforms'); tv.models = mod('torchvision.models')
    vggm = mod('torchvision.models.vgg'); tv.models.vgg = vggm; vggm.vgg16 = lambda **k: None
    og = mod('osgeo')
    gdal = mod('osgeo.gdal'); og.gdal = gdal
    gdal.Open = lambda p: _DS(SCENES[p]) if p in SCENES else None
    gdal.GDT_Float32 = 6
    og.ogr = mod('osgeo.ogr'); og.osr = mod('osgeo.osr')
    mod('cv2')


def main():
    install_stubs()
    sys.path.insert(0, REF)
    import data_utils as RD
    import metrics as RMET
    import CommonFunc as RC
    out = {}
    rng = np.random.default_rng(2025)
    cases = [('a', 4, 97, 131, (40, 40), (5, 5)), ('b', 3, 200, 200, (220, 220), (10, 10)),
             ('c', 2, 64, 50, (32, 24), (4, 3)), ('d', 1, 256, 256, (256, 256), (10, 10))]
    for tag, nb, ys, xs, patch, pad in cases:
        SCENES['x'] = rng.integers(0, 4000, (nb, ys, xs)).astype(np.uint16)
        SCENES['y'] = rng.integers(0, 4000, (nb, ys, xs)).astype(np.uint16)
        SCENES['r'] = rng.integers(1, 3, (1, ys, xs)).astype(np.uint8)
        mean_x, std_x = SCENES['x'].reshape(nb, -1).mean(1), SCENES['x'].reshape(nb, -1).std(1)
        mean_y, std_y = SCENES['y'].reshape(nb, -1).mean(1), SCENES['y'].reshape(nb, -1).std(1)
        ds = RD.GDALDataset('x', 'y', refPath='r', outPath='o', enhance=RC.NORMALIZE(mean_x, std_x, mean_y, std_y),
                            patch_size=patch, overlap_padding=pad)
        n = len(ds)
        sl = []
        for item in range(n):
            ix, iy = int(np.floor(item / ds.patch_count()[1])), item % ds.patch_count()[1]
            s, sr, sw = ds.slice_assign(ix, iy)
            sl.append(list(s) + list(sr) + list(sw))
        out[tag + '/meta'] = np.array([nb, ys, xs, patch[0], patch[1], pad[0], pad[1], n], np.int64)
        out[tag + '/x'] = SCENES['x']; out[tag + '/y'] = SCENES['y']; out[tag + '/r'] = SCENES['r']
        out[tag + '/stats'] = np.stack([mean_x, std_x, mean_y, std_y])
        out[tag + '/slices'] = np.array(sl, np.int64)
        out[tag + '/counts'] = np.array(ds.patch_count(), np.int64)
        pick = sorted(set([0, n - 1, n // 2, min(1, n - 1)]))
        out[tag + '/pick'] = np.array(pick, np.int64)
        for item in pick:
            px, py, it, pr = ds[item]
            out['%s/item%d/x' % (tag, item)] = px.numpy()
            out['%s/item%d/y' % (tag, item)] = py.numpy()
            out['%s/item%d/ref' % (tag, item)] = pr.numpy()
        # write-back: every tile writes (item + patch mean of band 0) into its centre
        for item in range(n):
            px, py, it, pr = ds[item]
            res = torch.full((1, patch[1], patch[0]), float(item)) + px[0:1] * 0 + px[0:1].mean()
            ds.GDALwrite(res, item)
        out[tag + '/written'] = SCENES['o'].copy()
    # Evaluator + colour coding
    gt = rng.integers(1, 3, (3, 50, 60)); pre = rng.integers(0, 2, (3, 50, 60))
    ev = RMET.Evaluator(num_class=2)
    for i in range(3):
        ev.add_batch_map(gt[i].astype(np.int16), pre[i].astype(np.int16), [1, 2], [0, 1])
    out['metrics/gt'] = gt; out['metrics/pre'] = pre
    out['metrics/cm'] = ev.confusion_matrix
    out['metrics/scores'] = np.array([ev.Pixel_Accuracy(), ev.Pixel_Kappa(), ev.Pixel_Precision_Rate(),
                                      ev.Pixel_Recall_Rate(), ev.Pixel_F1_score(),
                                      ev.Mean_Intersection_over_Union()[0], ev.Mean_Intersection_over_Union()[1],
                                      ev.Frequency_Weighted_Intersection_over_Union(), ev.Pixel_Accuracy_Class()[0]])
    out['metrics/codes_color'] = RC.write_changemap_gdal(pre[0:1], gt[0:1], True, ref_map=[1, 2], dt_map=[0, 1])
    out['metrics/codes_plain'] = RC.write_changemap_gdal(pre[0:1], gt[0:1], False, ref_map=[1, 2], dt_map=[0, 1])
    # LR schedule table
    class O:
        param_groups = [{'lr': 0}]
    lrs = []
    for ep in range(0, 40):
        o = O(); RC.adjust_learning_rate(o, ep, lr_start=1e-4, lr_max=1e-3, lr_warm_up_epoch=5); a = o.param_groups[0]['lr']
        o = O(); RC.adjust_learning_rate(o, ep, lr_start=1e-5, lr_max=3e-4, lr_warm_up_epoch=10, lr_sustain_epochs=10)
        lrs.append([a, o.param_groups[0]['lr']])
    out['lr/table'] = np.array(lrs)
    np.savez_compressed(os.path.join(HERE, 'tiles.npz'), **out)
    print('wrote tiles.npz', len(out), 'arrays')


if __name__ == '__main__':
    main()
