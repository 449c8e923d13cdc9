"""datasets.py vs fixtures from the reference's GDALDataset_RSS / OSCD_Dataset_RSS /
WHU_Dataset_WSS.order_reset (tests/golden/gen_golden_datasets.py).  Bit-exact."""
import os

import numpy as np
import pytest

from fcd_gan_pytorch_amd import datasets

G = os.path.join(os.path.dirname(__file__), 'golden', 'datasets.npz')


def build():
    z = np.load(G)
    scenes, names = [], ['abudhabi', 'paris']
    for nm in names:
        xname = str(z['%s/xname' % nm])
        yname = nm + ('_2' if xname.endswith('_1') else '_1')
        scenes.append(datasets.RegionTileDataset(z['%s/%s' % (nm, xname)], z['%s/%s' % (nm, yname)],
                                                 region=z['%s/%s-region.tif' % (nm, nm)],
                                                 ref=z['%s/%s-cm.tif' % (nm, nm)], patch_size=(40, 32),
                                                 overlap_padding=(4, 3)))
    return z, datasets.MultiSceneDataset(scenes, names)


def test_multi_scene_region_dataset_bit_exact():
    z, ds = build()
    assert [len(ds)] + list(ds.cumlen) == list(z['len'])
    np.testing.assert_array_equal(np.array([list(ds.eff_range(i)) for i in range(len(ds))]), z['eff'])
    for it in z['pick']:
        x, y, item, ref, region = ds[int(it)]
        assert int(item) == int(z['item%d/item' % it])
        for got, key in ((x, 'x'), (y, 'y'), (ref, 'ref'), (region, 'region')):
            np.testing.assert_array_equal(got.numpy(), z['item%d/%s' % (it, key)])
        assert set(np.unique(region.numpy())).issubset({0.0, 1.0})
    with pytest.raises(IndexError):
        ds[len(ds)]
    outs = ds.new_outputs()
    for it in range(len(ds)):
        ds.write_center(outs, np.full((1, 32, 40), float(it + 1), np.float32), it)
    assert all((o > 0).all() for o in outs)            # the owned centres tile every scene completely


@pytest.mark.parametrize('tag', ['p1', 'p2', 'p3'])
def test_pairing_order_matches_reference(tag):
    z = np.load(G)
    c, nc, seed = [int(v) for v in z[tag + '/meta']]
    p = datasets.PairingDataset(list(range(c)), list(range(100, 100 + nc)), random_assign=False, seed=0)
    p.order_reset(seed=seed)
    np.testing.assert_array_equal(np.array(p.cds_order), z[tag + '/cds'])
    np.testing.assert_array_equal(np.array(p.ncds_order), z[tag + '/ncds'])
    assert len(p) == max(c, nc)
    a, b = p[0]
    assert a == p.cds_order[0] and b == 100 + p.ncds_order[0]
