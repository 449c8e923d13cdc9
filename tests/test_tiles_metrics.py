"""Data-side rows (SURVEY 8f-1,2,4) vs fixtures produced by the reference's own GDALDataset /
Evaluator / write_changemap_gdal / NORMALIZE / adjust_learning_rate
(tests/golden/gen_golden_tiles.py), plus TIFF codec round trips (checked against Pillow where
Pillow can read the flavour).  Integer / index work is compared bit-exactly."""
import os

import numpy as np
import pytest
import torch

from fcd_gan_pytorch_amd import tiles, metrics
from fcd_gan_pytorch_amd.optim import adjust_learning_rate

G = os.path.join(os.path.dirname(__file__), 'golden', 'tiles.npz')


@pytest.mark.parametrize('tag', ['a', 'b', 'c', 'd'])
def test_tile_geometry_patches_and_writeback_bit_exact(tag):
    z = np.load(G)
    nb, ys, xs, pw, ph, padx, pady, n = [int(v) for v in z[tag + '/meta']]
    grid = tiles.TileGrid(xs, ys, (pw, ph), (padx, pady))
    assert len(grid) == n and list(grid.patch_count()) == list(z[tag + '/counts'])
    got = np.array([sum((list(t) for t in grid.slices(i)), []) for i in range(n)])
    np.testing.assert_array_equal(got, z[tag + '/slices'])
    st = z[tag + '/stats']
    ds = tiles.PairTileDataset(z[tag + '/x'], z[tag + '/y'], z[tag + '/r'], (pw, ph), (padx, pady),
                               stats=(st[0], st[1], st[2], st[3]))
    for item in z[tag + '/pick']:
        x, y, it, ref = ds[int(item)]
        assert int(it) == int(item)
        np.testing.assert_array_equal(x.numpy(), z['%s/item%d/x' % (tag, item)])
        np.testing.assert_array_equal(y.numpy(), z['%s/item%d/y' % (tag, item)])
        np.testing.assert_array_equal(ref.numpy(), z['%s/item%d/ref' % (tag, item)])
    out = np.zeros((1, ys, xs), np.float32)
    for item in range(n):
        x, _, _, _ = ds[item]
        res = torch.full((1, ph, pw), float(item)) + x[0:1] * 0 + x[0:1].mean()
        grid.write_center(out, res.numpy(), item)
        r0, r1, c0, c1 = grid.eff_range(item)
        (sx, sy, sw, sh), _, _ = grid.slices(item)
        assert (r1 - r0, c1 - c0) == (sh, sw)
    np.testing.assert_array_equal(out, z[tag + '/written'])


def test_tile_grid_rejects_bad_geometry():
    with pytest.raises(ValueError):
        tiles.TileGrid(100, 100, (20, 20), (10, 10))
    with pytest.raises(ValueError):
        tiles.PairTileDataset(np.zeros((2, 8, 8)), np.zeros((2, 8, 9)))


@pytest.mark.parametrize('dtype,bands,planar', [(np.uint8, 3, False), (np.uint8, 1, True), (np.uint16, 4, True),
                                                (np.uint16, 13, False), (np.float32, 1, True), (np.float32, 4, True)])
def test_tiff_round_trip_and_pillow(tmp_path, dtype, bands, planar):
    rng = np.random.default_rng(bands)
    a = (rng.uniform(0, 250, (bands, 37, 53))).astype(dtype)
    p = str(tmp_path / 't.tif')
    tiles.write_tiff(p, a, planar=planar, rows_per_strip=8)
    b = tiles.read_tiff(p)
    assert b.dtype == a.dtype and b.shape == a.shape
    np.testing.assert_array_equal(a, b)
    from PIL import Image
    if bands == 1 or (bands == 3 and dtype == np.uint8 and not planar):
        im = np.array(Image.open(p))
        ref = a[0] if bands == 1 else a.transpose(1, 2, 0)
        np.testing.assert_array_equal(im, ref)


def test_tiff_reads_pillow_written_and_rejects_compressed(tmp_path):
    from PIL import Image
    a = (np.arange(40 * 30) % 251).astype(np.uint8).reshape(40, 30)
    p = str(tmp_path / 'p.tif')
    Image.fromarray(a).save(p)
    np.testing.assert_array_equal(tiles.read_tiff(p)[0], a)
    f = np.linspace(-1, 1, 40 * 30, dtype=np.float32).reshape(40, 30)
    Image.fromarray(f).save(p)
    np.testing.assert_array_equal(tiles.read_tiff(p)[0], f)
    Image.fromarray(a).save(p, compression='tiff_lzw')
    with pytest.raises(ValueError):
        tiles.read_tiff(p)
    with open(p, 'wb') as fh:
        fh.write(b'not a tiff')
    with pytest.raises(ValueError):
        tiles.read_tiff(p)


def test_evaluator_and_colour_codes_match_reference():
    z = np.load(G)
    gt, pre = torch.from_numpy(z['metrics/gt']), torch.from_numpy(z['metrics/pre'])
    ev = metrics.Evaluator(2)
    ev.add_batch_map(gt, pre, [1, 2], [0, 1])            # whole batch at once
    np.testing.assert_array_equal(ev._sync(), z['metrics/cm'])
    got = np.array([ev.Pixel_Accuracy(), ev.Pixel_Kappa(), ev.Pixel_Precision_Rate(), ev.Pixel_Recall_Rate(),
                    ev.Pixel_F1_score(), ev.Mean_Intersection_over_Union()[0], ev.Mean_Intersection_over_Union()[1],
                    ev.Frequency_Weighted_Intersection_over_Union(), ev.Pixel_Accuracy_Class()[0]])
    np.testing.assert_array_equal(got, z['metrics/scores'])
    c = metrics.changemap_codes(pre[0:1], gt[0:1], True, ref_map=(1, 2), dt_map=(0, 1))
    np.testing.assert_array_equal(c.numpy(), z['metrics/codes_color'])
    c = metrics.changemap_codes(pre[0:1], gt[0:1], False, ref_map=(1, 2), dt_map=(0, 1))
    np.testing.assert_array_equal(c.numpy(), z['metrics/codes_plain'])
    ev.reset()
    valid = torch.zeros(gt.shape, dtype=torch.bool)
    valid[:, 5:20, 7:33] = True
    ev.add_batch_map(gt, pre, [1, 2], [0, 1], valid=valid)
    assert ev._sync().sum() == 3 * 15 * 26
    assert torch.equal(metrics.threshold_map(torch.tensor([0.2, 0.5, 0.7])), torch.tensor([0., 0., 1.]))


def test_lr_schedule_table_matches_reference():
    z = np.load(G)

    class O:
        param_groups = [{'lr': 0}]
    got = []
    for ep in range(40):
        o = O()
        a = adjust_learning_rate(o, ep, lr_start=1e-4, lr_max=1e-3, lr_warm_up_epoch=5)
        b = adjust_learning_rate(o, ep, lr_start=1e-5, lr_max=3e-4, lr_warm_up_epoch=10, lr_sustain_epochs=10)
        got.append([a, b])
    np.testing.assert_array_equal(np.array(got), z['lr/table'])


def test_prefetcher_yields_batches_in_order_on_cpu():
    data = [(torch.full((2, 3), float(i)), torch.tensor(i)) for i in range(5)]
    out = list(tiles.Prefetcher(data, 'cpu'))
    assert [int(b[1]) for b in out] == list(range(5))
    assert all(torch.equal(o[0], d[0]) for o, d in zip(out, data))


@pytest.mark.parametrize('tag', ['s1', 's2'])
def test_dataset_statistics_match_reference(tag, tmp_path):
    """Dataset_mean / Dataset_std / Dataset_meanstd (valid-pixel mask from the first scene,
    count-weighted per-patch moments, text-file cache) vs the reference run over the same scenes
    (tests/golden/gen_golden_stats.py)."""
    from fcd_gan_pytorch_amd import tiles
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'stats.npz'))
    nb, ys, xs, pw, ph, ox, oy = [int(v) for v in z[tag + '/meta']]
    ds = tiles.PairTileDataset(z[tag + '/x'], z[tag + '/y'], None, (pw, ph), (ox, oy))
    mx, my = tiles.dataset_mean(ds)
    sx, sy = tiles.dataset_std(ds, mx, my)
    ref = z[tag + '/mean_std']
    np.testing.assert_allclose(np.stack([mx.numpy(), sx.numpy(), my.numpy(), sy.numpy()]), ref, rtol=1e-6)
    t1, t2 = str(tmp_path / 'x.txt'), str(tmp_path / 'y.txt')
    first = tiles.dataset_meanstd(t1, t2, ds)
    np.testing.assert_allclose(np.array(first), z[tag + '/first'], rtol=1e-6)
    # same text format as the reference (two lines, "mean:" / "std:" + space separated values)
    for path, key in ((t1, '/txt1'), (t2, '/txt2')):
        mine, theirs = open(path).read().split(), bytes(z[tag + key]).decode().split()
        assert [w for w in mine if w.endswith(':')] == [w for w in theirs if w.endswith(':')] == ['mean:', 'std:']
        np.testing.assert_allclose([float(w) for w in mine if not w.endswith(':')],
                                   [float(w) for w in theirs if not w.endswith(':')], rtol=1e-6)
    again = tiles.dataset_meanstd(t1, t2, ds)           # second call reads the cache
    np.testing.assert_allclose(np.array(again), np.array(first), rtol=1e-7)
    np.testing.assert_allclose(np.array(again), z[tag + '/again'], rtol=1e-6)
    # the reference's own cache files are read back identically
    open(t1, 'wb').write(bytes(z[tag + '/txt1'])); open(t2, 'wb').write(bytes(z[tag + '/txt2']))
    np.testing.assert_allclose(np.array(tiles.dataset_meanstd(t1, t2, ds)), z[tag + '/again'], rtol=0, atol=0)
