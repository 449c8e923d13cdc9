"""Dataset compositions used by the RSSS / WSSS demos, restated over ``tiles.PairTileDataset``
(in-memory scenes or TIFF paths; GDAL / the OSCD and WHU directory conventions are out of
scope -- SURVEY.md section 2 -- only the tuple layouts and index arithmetic the train steps
depend on are kept):

 * ``RegionTileDataset``  -- GDALDataset_RSS (data_utils.py:239-290): adds the weak *region*
   mask, binarised with ``> 125 -> 1``; tuple ``(x, y, item, ref, region)``.
 * ``MultiSceneDataset``  -- OSCD_Dataset_RSS (data_utils.py:294-446): several scenes
   concatenated, global item index, ``EffRange`` and per-scene centre write-back.
 * ``PairingDataset``     -- WHU_Dataset_WSS (data_utils.py:570-625): pairs every sample of the
   larger of {changed, unchanged} with a (re-shuffled, repeated) sample of the smaller one;
   ``order_reset`` is called once per epoch (Demo_WSSS.py:247) and must use the same seed on
   every rank under data parallelism.
"""
import math
import random

import numpy as np
import torch

from . import tiles


class RegionTileDataset(tiles.PairTileDataset):
    def __init__(self, scene_x, scene_y, region=None, ref=None, patch_size=(200, 200), overlap_padding=(10, 10),
                 stats=None):
        super().__init__(scene_x, scene_y, ref, patch_size, overlap_padding, stats)
        if isinstance(region, str):
            region = tiles.read_tiff(region)
        if region is not None and (region.shape[0] != 1 or region.shape[1:] != self.x.shape[1:]):
            raise ValueError("Reference sizes don't match image")
        self.region = region

    def __getitem__(self, item):
        x, y, it, ref = super().__getitem__(item)
        _, (rx, ry, rw, rh), (wx, wy, ww, wh) = self.grid.slices(item)
        px, py = self.grid.patch_size
        reg = np.zeros((1, py, px), dtype=float)
        if self.region is not None:
            reg[:, wy:wy + wh, wx:wx + ww] = self.region[:, ry:ry + rh, rx:rx + rw]
        reg[reg > 125] = 1                                   # data_utils.py:280
        return x, y, it, ref, torch.from_numpy(reg).float()


class MultiSceneDataset(torch.utils.data.Dataset):
    """Concatenation of per-scene tile datasets with a global item index."""

    def __init__(self, scenes, names=None):
        self.scenes = list(scenes)
        self.names = list(names) if names is not None else ['scene%d' % i for i in range(len(self.scenes))]
        self.numlist = [len(s) for s in self.scenes]
        self.cumlen = np.cumsum(np.array(self.numlist)).tolist()

    def __len__(self):
        return int(sum(self.numlist))

    def locate(self, item):
        """global item -> (scene index, item within the scene)  (data_utils.py:375-376)."""
        item = int(item)
        if item >= self.cumlen[-1] or item < 0:
            raise IndexError('item exceeds the len')
        ds = int(np.where(np.array(self.cumlen) > item)[0][0])
        return ds, (item - self.cumlen[ds - 1] if ds > 0 else item)

    def __getitem__(self, item):
        ds, cur = self.locate(item)
        out = list(self.scenes[ds][cur])
        out[2] = out[2] + self.cumlen[ds - 1] if ds > 0 else out[2]
        return tuple(out)

    def eff_range(self, item):
        ds, cur = self.locate(item)
        return self.scenes[ds].grid.eff_range(cur)

    def new_outputs(self, bands=1, dtype=np.float32):
        return [np.zeros((bands, s.grid.ysize, s.grid.xsize), dtype) for s in self.scenes]

    def write_center(self, outputs, patch, item):
        ds, cur = self.locate(item)
        self.scenes[ds].grid.write_center(outputs[ds], patch, cur)


class PairingDataset(torch.utils.data.Dataset):
    """(changed sample, unchanged sample) pairs -- WHU_Dataset_WSS."""

    def __init__(self, changed, unchanged, random_assign=True, seed=None):
        self.cDS, self.ncDS = changed, unchanged
        self.cds_len, self.ncds_len = len(changed), len(unchanged)
        self.random_assign = random_assign
        self.rng = random.Random(seed) if seed is not None else random
        if not random_assign:
            self.order_reset()

    def order_reset(self, seed=None):
        """Repeat + reshuffle the smaller class to the length of the larger one
        (data_utils.py:586-605).  Pass the same ``seed`` on every rank."""
        rng = random.Random(seed) if seed is not None else self.rng
        big, small = (self.cds_len, self.ncds_len) if self.cds_len > self.ncds_len else (self.ncds_len, self.cds_len)
        order_temp = [i for i in range(small)]
        order = []
        for _ in range(math.ceil(big / small)):
            rng.shuffle(order_temp)
            order = order + order_temp
        order = order[:big]
        if self.cds_len > self.ncds_len:
            self.ncds_order, self.cds_order = order, [i for i in range(self.cds_len)]
        else:
            self.cds_order, self.ncds_order = order, [i for i in range(self.ncds_len)]

    def __getitem__(self, item):
        if not self.random_assign:
            item_ncds, item_cds = self.ncds_order[item], self.cds_order[item]
        elif self.cds_len > self.ncds_len:
            item_cds, item_ncds = item, self.rng.randint(0, self.ncds_len - 1)
        else:
            item_ncds, item_cds = item, self.rng.randint(0, self.cds_len - 1)
        return self.cDS[item_cds], self.ncDS[item_ncds]

    def __len__(self):
        return max(self.cds_len, self.ncds_len)
