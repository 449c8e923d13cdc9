"""Tile geometry, minimal TIFF I/O and host->device feeding for the hot path
(SURVEY.md section 8f rows 1 and 4).

The reference tiles a big GeoTIFF pair into overlapped patches through GDAL
(``GDALDataset``, data_utils.py:28-236): stride = patch - 2*pad, every patch is read
with ``pad`` pixels of context (clipped at the scene border), zero-embedded into a
fixed ``patch_size`` canvas, and only its centre is written back.  GDAL is not
available here, so this module restates the *arithmetic* (``TileGrid``), reads /
writes baseline TIFF itself (uncompressed strips; uint8 / uint16 / float32; chunky or
planar samples -- what config[0]'s synthetic 4-band T1/T2 pair needs) and feeds the GPU
from a background thread through pinned staging buffers (``Prefetcher``).

Per-band normalisation ``(x - mean) / std`` (``NORMALIZE``, CommonFunc.py:199-224) is
applied on the host copy exactly like the reference (``enhance``), or on device via
``normalize_``.
"""
import math
import struct
import threading
import queue

import os

import numpy as np
import torch


# ----------------------------------------------------------------------- geometry
class TileGrid:
    """Overlapped tiling of an (xsize, ysize) scene -- data_utils.py:57-70,142-176.
    ``patch_size`` and ``overlap_padding`` are (x, y) pairs like the reference's."""

    def __init__(self, xsize, ysize, patch_size=(200, 200), overlap_padding=(10, 10)):
        self.xsize, self.ysize = int(xsize), int(ysize)
        self.patch_size = (int(patch_size[0]), int(patch_size[1]))
        self.pad = (int(overlap_padding[0]), int(overlap_padding[1]))
        sx = self.patch_size[0] - 2 * self.pad[0]
        sy = self.patch_size[1] - 2 * self.pad[1]
        if sx <= 0 or sy <= 0:
            raise ValueError('patch_size must exceed twice the overlap padding')
        self.xstart = list(range(0, self.xsize, sx))
        self.xend = [x + sx for x in self.xstart if x + sx < self.xsize] + [self.xsize]
        self.ystart = list(range(0, self.ysize, sy))
        self.yend = [y + sy for y in self.ystart if y + sy < self.ysize] + [self.ysize]

    def patch_count(self):
        return len(self.xstart), len(self.ystart)

    def __len__(self):
        return len(self.xstart) * len(self.ystart)

    def item_xy(self, item):
        ny = len(self.ystart)
        return int(math.floor(item / ny)), int(item % ny)

    def slices(self, item):
        """(slice, slice_read, slice_write), each (x0, y0, w, h):
        slice = the centre region this tile owns in the scene; slice_read = what is read
        (centre + clipped context); slice_write = where the read block sits in the canvas."""
        ix, iy = self.item_xy(item)
        px, py = self.pad
        x0, x1, y0, y1 = self.xstart[ix], self.xend[ix], self.ystart[iy], self.yend[iy]
        own = (x0, y0, x1 - x0, y1 - y0)
        ox = 0 if x0 - px > 0 else px
        oy = 0 if y0 - py > 0 else py
        rx0 = x0 - px if x0 - px > 0 else 0
        ry0 = y0 - py if y0 - py > 0 else 0
        rx1 = x1 + px if x1 + px < self.xsize else self.xsize
        ry1 = y1 + py if y1 + py < self.ysize else self.ysize
        return own, (rx0, ry0, rx1 - rx0, ry1 - ry0), (ox, oy, rx1 - rx0, ry1 - ry0)

    def read_patch(self, scene, item, dtype=np.float64):
        """scene: (bands, ysize, xsize) array -> (bands, patch_y, patch_x) zero-embedded patch
        (data_utils.py:98-118)."""
        _, (rx, ry, rw, rh), (wx, wy, ww, wh) = self.slices(item)
        canvas = np.zeros((scene.shape[0], self.patch_size[1], self.patch_size[0]), dtype=dtype)
        canvas[:, wy:wy + wh, wx:wx + ww] = scene[:, ry:ry + rh, rx:rx + rw]
        return canvas

    def write_center(self, out_scene, patch, item):
        """Write only the owned centre of a (bands, patch_y, patch_x) result back into
        out_scene (bands, ysize, xsize) -- data_utils.py:205-213,230-236."""
        (x0, y0, w, h), _, _ = self.slices(item)
        px, py = self.pad
        out_scene[:, y0:y0 + h, x0:x0 + w] = patch[:, py:py + h, px:px + w]

    def eff_range(self, item):
        """Rows/cols of a patch that are scored by the metrics (OSCD_Dataset_RSS.EffRange,
        data_utils.py:390-400 semantics: the owned centre inside the canvas)."""
        (x0, y0, w, h), _, _ = self.slices(item)
        px, py = self.pad
        return py, py + h, px, px + w


# --------------------------------------------------------------------- TIFF codec
_TIFF_TYPES = {1: ('B', 1), 2: ('c', 1), 3: ('H', 2), 4: ('I', 4), 5: ('II', 8), 16: ('Q', 8)}
_NP_OF = {(1, 8): np.uint8, (1, 16): np.uint16, (1, 32): np.uint32, (2, 16): np.int16, (2, 32): np.int32,
          (3, 32): np.float32, (3, 64): np.float64}


def read_tiff(path):
    """Baseline TIFF reader: uncompressed, strip-organised, chunky or planar samples.
    Returns a (bands, height, width) array in the file's sample dtype."""
    with open(path, 'rb') as f:
        data = f.read()
    bo = {b'II': '<', b'MM': '>'}.get(data[:2])
    if bo is None or struct.unpack(bo + 'H', data[2:4])[0] != 42:
        raise ValueError('%s: not a classic TIFF file' % path)
    off = struct.unpack(bo + 'I', data[4:8])[0]
    n = struct.unpack(bo + 'H', data[off:off + 2])[0]
    tags = {}
    for i in range(n):
        e = off + 2 + 12 * i
        tag, typ, cnt = struct.unpack(bo + 'HHI', data[e:e + 8])
        if typ not in _TIFF_TYPES:
            continue
        code, size = _TIFF_TYPES[typ]
        total = size * cnt
        voff = e + 8 if total <= 4 else struct.unpack(bo + 'I', data[e + 8:e + 12])[0]
        if typ == 5:
            vals = struct.unpack(bo + 'I' * (2 * cnt), data[voff:voff + total])
        elif typ == 2:
            vals = (data[voff:voff + cnt],)
        else:
            vals = struct.unpack(bo + code * cnt, data[voff:voff + total])
        tags[tag] = vals
    W, H = tags[256][0], tags[257][0]
    spp = tags.get(277, (1,))[0]
    bits = tags.get(258, (1,) * spp)
    if tags.get(259, (1,))[0] != 1:
        raise ValueError('%s: compressed TIFF is not supported' % path)
    if 324 in tags:
        raise ValueError('%s: tiled TIFF is not supported' % path)
    fmt = tags.get(339, (1,) * spp)[0]
    planar = tags.get(284, (1,))[0]
    dt = _NP_OF.get((fmt, bits[0]))
    if dt is None or any(b != bits[0] for b in bits):
        raise ValueError('%s: unsupported sample format %r / bits %r' % (path, fmt, bits))
    dt = np.dtype(dt).newbyteorder(bo)
    offs, cnts = tags[273], tags[279]
    raw = b''.join(data[o:o + c] for o, c in zip(offs, cnts))
    arr = np.frombuffer(raw, dtype=dt)
    if planar == 2:
        out = arr[:spp * H * W].reshape(spp, H, W)
    else:
        out = arr[:H * W * spp].reshape(H, W, spp).transpose(2, 0, 1)
    return np.ascontiguousarray(out).astype(dt.newbyteorder('='))


def write_tiff(path, array, planar=True, rows_per_strip=None):
    """Write a (bands, H, W) (or (H, W)) uint8 / uint16 / float32 array as baseline TIFF
    (little-endian, uncompressed strips)."""
    a = np.asarray(array)
    if a.ndim == 2:
        a = a[None]
    fmt_bits = {np.dtype(np.uint8): (1, 8), np.dtype(np.uint16): (1, 16), np.dtype(np.float32): (3, 32)}.get(a.dtype)
    if fmt_bits is None:
        raise ValueError('write_tiff supports uint8, uint16 and float32, got %s' % a.dtype)
    fmt, bits = fmt_bits
    spp, H, W = a.shape
    rps = H if rows_per_strip is None else int(rows_per_strip)
    nstrips_plane = (H + rps - 1) // rps
    if planar and spp > 1:
        payload = [np.ascontiguousarray(a[b, r:r + rps]).astype('<' + a.dtype.str[1:]).tobytes()
                   for b in range(spp) for r in range(0, H, rps)]
        planar_cfg = 2
    else:
        chunky = np.ascontiguousarray(a.transpose(1, 2, 0))
        payload = [chunky[r:r + rps].astype('<' + a.dtype.str[1:]).tobytes() for r in range(0, H, rps)]
        planar_cfg = 1
    nstr = len(payload)
    assert nstr == nstrips_plane * (spp if planar_cfg == 2 else 1)
    entries = []     # (tag, type, count, values)

    def ent(tag, typ, vals):
        entries.append((tag, typ, len(vals), list(vals)))
    ent(256, 4, [W]); ent(257, 4, [H]); ent(258, 3, [bits] * spp); ent(259, 3, [1])
    rgb = spp == 3 and a.dtype == np.uint8           # like GDAL: 3 x uint8 is written as RGB
    ent(262, 3, [2 if rgb else 1]); ent(273, 4, [0] * nstr); ent(277, 3, [spp]); ent(278, 4, [rps])
    ent(279, 4, [len(p) for p in payload]); ent(284, 3, [planar_cfg]); ent(339, 3, [fmt] * spp)
    if spp > 1 and not rgb:
        ent(338, 3, [0] * (spp - 1))          # ExtraSamples: unspecified
    entries.sort(key=lambda e: e[0])
    ifd_off = 8
    ifd_size = 2 + 12 * len(entries) + 4
    extra_off = ifd_off + ifd_size
    extra = b''
    recs = []
    strip_slot = None
    for tag, typ, cnt, vals in entries:
        code = {3: 'H', 4: 'I'}[typ]
        blob = struct.pack('<' + code * cnt, *vals)
        if len(blob) <= 4:
            recs.append([tag, typ, cnt, blob.ljust(4, b'\0'), None])
        else:
            recs.append([tag, typ, cnt, struct.pack('<I', extra_off + len(extra)), len(extra)])
            if tag == 273:
                strip_slot = (len(extra), cnt)
            extra += blob + (b'\0' if len(blob) % 2 else b'')
        if tag == 273 and len(blob) <= 4:
            strip_slot = ('inline', len(recs) - 1)
    data_off = extra_off + len(extra)
    offs, o = [], data_off
    for p in payload:
        offs.append(o)
        o += len(p)
    blob = struct.pack('<' + 'I' * nstr, *offs)
    if strip_slot[0] == 'inline':
        recs[strip_slot[1]][3] = blob.ljust(4, b'\0')
    else:
        extra = extra[:strip_slot[0]] + blob + extra[strip_slot[0] + len(blob):]
    with open(path, 'wb') as f:
        f.write(b'II' + struct.pack('<HI', 42, ifd_off))
        f.write(struct.pack('<H', len(recs)))
        for tag, typ, cnt, val, _ in recs:
            f.write(struct.pack('<HHI', tag, typ, cnt) + val)
        f.write(struct.pack('<I', 0))
        f.write(extra)
        for p in payload:
            f.write(p)


# ------------------------------------------------------------------------ dataset
class PairTileDataset(torch.utils.data.Dataset):
    """In-memory restatement of GDALDataset (data_utils.py:28-140): a bi-temporal scene pair
    (+ optional reference map) cut into overlapped patches.  ``__getitem__`` returns the
    reference's tuple ``(x, y, item, ref)`` of float32 tensors (data_utils.py:140)."""

    def __init__(self, scene_x, scene_y, ref=None, patch_size=(200, 200), overlap_padding=(10, 10),
                 stats=None):
        if isinstance(scene_x, str):
            scene_x = read_tiff(scene_x)
        if isinstance(scene_y, str):
            scene_y = read_tiff(scene_y)
        if isinstance(ref, str):
            ref = read_tiff(ref)
        if scene_x.shape != scene_y.shape:
            raise ValueError("Image sizes don't match")
        if ref is not None and (ref.shape[0] != 1 or ref.shape[1:] != scene_x.shape[1:]):
            raise ValueError("Reference sizes don't match image")
        self.x, self.y, self.ref = scene_x, scene_y, ref
        self.grid = TileGrid(scene_x.shape[2], scene_x.shape[1], patch_size, overlap_padding)
        self.stats = stats            # (meanX, stdX, meanY, stdY) per band, NORMALIZE semantics

    def __len__(self):
        return len(self.grid)

    def _norm(self, a, mean, std):
        a = a.astype(float)
        for b in range(a.shape[0]):
            a[b] = (a[b] - mean[b]) / std[b]
        return a

    def raw_item(self, item):
        """The patch WITHOUT host normalisation: ``(x_raw, y_raw, item, ref, valid)`` with ``valid`` (1,py,px) = 1 on
        the window that holds scene pixels.  Normalise on the device (``_ops.normalize_tiles``: one fused pass, fp64 per
        element, bit-identical to ``__getitem__``) or not at all (``Segmentor.forward_raw`` folds the statistics into
        its first convolution)."""
        _, (rx, ry, rw, rh), (wx, wy, ww, wh) = self.grid.slices(item)
        px, py = self.grid.patch_size
        cx = np.zeros((self.x.shape[0], py, px), dtype=np.float32)
        cy = np.zeros_like(cx)
        cx[:, wy:wy + wh, wx:wx + ww] = self.x[:, ry:ry + rh, rx:rx + rw]
        cy[:, wy:wy + wh, wx:wx + ww] = self.y[:, ry:ry + rh, rx:rx + rw]
        r = np.zeros((1, py, px), dtype=np.float32)
        if self.ref is not None:
            r[:, wy:wy + wh, wx:wx + ww] = self.ref[:, ry:ry + rh, rx:rx + rw]
        valid = np.zeros((1, py, px), dtype=np.float32)
        valid[:, wy:wy + wh, wx:wx + ww] = 1.0
        return (torch.from_numpy(cx), torch.from_numpy(cy), torch.tensor(item), torch.from_numpy(r),
                torch.from_numpy(valid))

    def __getitem__(self, item):
        _, (rx, ry, rw, rh), (wx, wy, ww, wh) = self.grid.slices(item)
        bx = np.array(self.x[:, ry:ry + rh, rx:rx + rw], dtype=float)
        by = np.array(self.y[:, ry:ry + rh, rx:rx + rw], dtype=float)
        if self.stats is not None:                      # 'enhance' runs on the read block (data_utils.py:106-108)
            bx = self._norm(bx, self.stats[0], self.stats[1])
            by = self._norm(by, self.stats[2], self.stats[3])
        px, py = self.grid.patch_size
        cx = np.zeros((bx.shape[0], py, px), dtype=float)
        cy = np.zeros_like(cx)
        cx[:, wy:wy + wh, wx:wx + ww] = bx
        cy[:, wy:wy + wh, wx:wx + ww] = by
        r = np.zeros((1, py, px), dtype=float)
        if self.ref is not None:
            r[:, wy:wy + wh, wx:wx + ww] = self.ref[:, ry:ry + rh, rx:rx + rw]
        return (torch.from_numpy(cx).float(), torch.from_numpy(cy).float(), torch.tensor(item),
                torch.from_numpy(r).float())


class RawTiles(torch.utils.data.Dataset):
    """View of a PairTileDataset that yields ``raw_item`` tuples (normalisation left to the device)."""

    def __init__(self, ds):
        self.ds, self.grid, self.ref, self.stats = ds, ds.grid, ds.ref, ds.stats

    def __len__(self):
        return len(self.ds)

    def __getitem__(self, item):
        return self.ds.raw_item(item)


# ------------------------------------------------------------------------ dataset statistics
def _valid(x):
    """Valid-pixel mask of a patch: pixels whose band sum is non-zero (nodata / zero padding excluded,
    CommonFunc.py:446) -- taken from the FIRST scene for both scenes, as the reference does."""
    return torch.sum(x, dim=0) != 0


def dataset_mean(dataset):
    """Per-band mean of both scenes over the valid pixels of all patches (CommonFunc.Dataset_mean,
    CommonFunc.py:436-465): per-patch means weighted by the patch's valid-pixel count."""
    npix, mx, my = [], [], []
    for i in range(len(dataset)):
        item = dataset[i]
        x, y = item[0], item[1]
        idx = _valid(x)
        npix.append(torch.sum(idx))
        mx.append(torch.mean(x[:, idx], 1))
        my.append(torch.mean(y[:, idx], 1))
    mx = torch.cat(mx).reshape(len(mx), -1)
    my = torch.cat(my).reshape(len(my), -1)
    npix = torch.tensor(npix).reshape(-1, 1)
    w = npix.repeat(1, mx.size()[1]) / torch.sum(npix)
    return torch.sum(mx * w, dim=0), torch.sum(my * w, dim=0)


def dataset_std(dataset, mean_x, mean_y):
    """Per-band standard deviation around the given means (CommonFunc.Dataset_std, CommonFunc.py:467-500):
    per-patch mean squared deviations weighted by n_i / (N - 1)."""
    npix, vx, vy = [], [], []
    mean_x = mean_x.reshape(-1, 1)
    mean_y = mean_y.reshape(-1, 1)
    for i in range(len(dataset)):
        item = dataset[i]
        x, y = item[0], item[1]
        idx = _valid(x)
        n = torch.sum(idx)
        npix.append(n)
        vx.append(torch.mean(torch.square(x[:, idx] - mean_x.repeat(1, n)), 1))
        vy.append(torch.mean(torch.square(y[:, idx] - mean_y.repeat(1, n)), 1))
    vx = torch.cat(vx).reshape(len(vx), -1)
    vy = torch.cat(vy).reshape(len(vy), -1)
    npix = torch.tensor(npix).reshape(-1, 1)
    w = npix.repeat(1, vx.size()[1]) / (torch.sum(npix) - 1)
    return torch.sqrt(torch.sum(vx * w, dim=0)), torch.sqrt(torch.sum(vy * w, dim=0))


def _write_stats(path, mean, std):
    with open(path, 'w') as f:
        f.write('mean:' + ''.join(' {}'.format(v) for v in mean) + '\n')
        f.write('std:' + ''.join(' {}'.format(v) for v in std) + '\n')


def _read_stats(path):
    with open(path) as f:
        lines = f.readlines()
    return [float(v) for v in lines[0].split()[1:]], [float(v) for v in lines[1].split()[1:]]


def dataset_meanstd(txt_x, txt_y, dataset):
    """``(meanX, stdX, meanY, stdY)`` as lists of floats, cached in two text files in the reference's
    format (``mean: v v ...`` / ``std: v v ...``; CommonFunc.Dataset_meanstd, CommonFunc.py:373-434):
    computed and written when either file is missing, read back otherwise."""
    if not (os.path.exists(txt_x) and os.path.exists(txt_y)):
        mx, my = dataset_mean(dataset)
        sx, sy = dataset_std(dataset, mx, my)
        _write_stats(txt_x, mx, sx)
        _write_stats(txt_y, my, sy)
        return mx.numpy().tolist(), sx.numpy().tolist(), my.numpy().tolist(), sy.numpy().tolist()
    mx, sx = _read_stats(txt_x)
    my, sy = _read_stats(txt_y)
    return mx, sx, my, sy


def normalize_(x, mean, std):
    """In-place per-band (x - mean) / std of an (N,C,H,W) device tensor (NORMALIZE on device)."""
    m = torch.as_tensor(mean, dtype=x.dtype, device=x.device).view(1, -1, 1, 1)
    s = torch.as_tensor(std, dtype=x.dtype, device=x.device).view(1, -1, 1, 1)
    return x.sub_(m).div_(s)


class Prefetcher:
    """Iterate a DataLoader-like iterable of tensor tuples from a background thread, staging
    each batch through pinned host memory and an asynchronous H2D copy on a side stream, so
    tile loading overlaps the train step (the reference loads on the main thread,
    num_workers=0, Demo_RSSS.py:242)."""

    def __init__(self, loader, device, depth=2):
        self.loader, self.device, self.depth = loader, torch.device(device), depth
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == 'cuda' else None

    def __iter__(self):
        q = queue.Queue(maxsize=self.depth)
        stop = object()
        failure = []

        def work():
            try:
                for batch in self.loader:
                    out = []
                    for t in batch:
                        if torch.is_tensor(t) and t.is_floating_point() and self.stream is not None:
                            p = t.pin_memory()
                            with torch.cuda.stream(self.stream):
                                d = p.to(self.device, non_blocking=True)
                            out.append((d, p))
                        else:
                            out.append((t, None))
                    ev = None
                    if self.stream is not None:
                        ev = torch.cuda.Event()
                        ev.record(self.stream)
                    q.put((out, ev))
            except BaseException as e:          # surfaced in the consumer: a dying loader must not end the epoch quietly
                failure.append(e)
            finally:
                q.put(stop)
        th = threading.Thread(target=work, daemon=True)
        th.start()
        while True:
            item = q.get()
            if item is stop:
                break
            out, ev = item
            if ev is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                for d, p in out:
                    if p is not None:
                        # the block was allocated on the copy stream: tell the caching allocator that the consumer
                        # stream uses it too, or it may be handed to the worker's next H2D copy while train-step
                        # kernels queued on the main stream still read it (the step functions never sync the host)
                        d.record_stream(cur)
            yield tuple(d for d, _ in out)
        th.join()
        if failure:
            raise failure[0]
