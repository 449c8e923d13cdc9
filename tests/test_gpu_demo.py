"""End-to-end plumbing of BASELINE.json configs[0] ("Demo_USSS on one synthetic T1/T2 .tif
pair") on the HIP path: TIFF pair -> overlapped tiles -> G pre-train / S pre-train / joint
epochs -> inference with centre write-back -> density TIFF + colour codes + metrics.
Parity: the stitched density map equals the CPU oracle run tile by tile with the trained
weights (<= 1e-4), the thresholded map bit-exactly outside the error margin."""
import numpy as np
import pytest
import torch

from oracle import nets as onets

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('H,W,patch,overlap', [(300, 330, 200, 10),      # ragged scene: 4 overlapped 200-px patches, padded edges
                                                (256, 256, 256, 0)])       # BASELINE.json configs[0] to the letter: ONE 256x256x4 .tif pair = one tile
def test_demo_usss_end_to_end(tmp_path, H, W, patch, overlap):
    from fcd_gan_pytorch_amd import demos, tiles
    rng = np.random.default_rng(11)
    C = 4
    t1 = rng.integers(100, 3000, (C, H, W)).astype(np.uint16)
    t2 = (t1 + rng.integers(-40, 40, (C, H, W))).clip(0, 65535).astype(np.uint16)
    c1 = min(300, W - 6)
    t2[:, 60:140, 200:c1] = rng.integers(100, 3000, (C, 80, c1 - 200))
    ref = np.ones((1, H, W), np.uint8)
    ref[:, 60:140, 200:c1] = 2
    px, py, pr = str(tmp_path / 'T1.tif'), str(tmp_path / 'T2.tif'), str(tmp_path / 'ref.tif')
    tiles.write_tiff(px, t1); tiles.write_tiff(py, t2); tiles.write_tiff(pr, ref)
    logs = []
    out = demos.demo_usss(px, py, pr, patch_size=(patch, patch), overlap_padding=(overlap, overlap), epochs_g=2, epochs_s=1,
                          epochs_joint=1, batch_size=2, allow_seeded=True, out_density=str(tmp_path / 'density.tif'),
                          out_color=str(tmp_path / 'color.tif'), log=logs.append)
    assert len(logs) == 4 and all(np.isfinite(v) for k in out['history'] for v in out['history'][k])
    assert out['history']['g'][1] < out['history']['g'][0]            # G pre-training reduces its loss
    dens = tiles.read_tiff(str(tmp_path / 'density.tif'))
    assert dens.dtype == np.float32 and dens.shape == (1, H, W)
    np.testing.assert_array_equal(dens, out['density'])
    assert 0.0 < dens.min() and dens.max() < 1.0
    # oracle: eval-mode forward per tile with the trained weights, stitched with the same geometry
    netS = out['netS']
    sd = {k: v.detach().cpu() for k, v in netS.state_dict().items()}
    stats = demos._tile_stats(t1, t2, (patch, patch))      # Dataset_meanstd over the non-overlapped tiling (Demo_USSS.py:88-95)
    ds = tiles.PairTileDataset(t1, t2, ref, (patch, patch), (overlap, overlap), stats=stats)
    assert len(ds) == (4 if patch == 200 else 1)
    want = np.zeros((1, H, W), np.float32)
    for item in range(len(ds)):
        x, y, _, _ = ds[item]
        o = onets.segmentor(sd, x[None], y[None], train=False, bilinear=True)[0].numpy()
        ds.grid.write_center(want, o, item)
    err = np.abs(dens - want).max()
    assert err <= 1e-4, err
    safe = np.abs(want - 0.5) > 2e-4
    assert np.array_equal((dens > 0.5)[safe], (want > 0.5)[safe])
    # metrics over the owned centres == whole scene exactly once
    cm = out['evaluator']._sync()
    assert cm.sum() == H * W
    pred = (dens > 0.5)[0]
    assert cm[1, 1] == np.sum(pred & (ref[0] == 2)) and cm[0, 1] == np.sum(pred & (ref[0] == 1))
    col = tiles.read_tiff(str(tmp_path / 'color.tif'))
    assert set(np.unique(col)).issubset({0.0, 1.0, 2.0, 3.0})


def _scene(rng, C, H, W):
    t1 = rng.standard_normal((C, H, W)).astype(np.float32)
    t2 = (t1 + 0.1 * rng.standard_normal((C, H, W))).astype(np.float32)
    r0, c0 = H // 4, W // 3
    t2[:, r0:r0 + 60, c0:c0 + 70] = rng.standard_normal((C, 60, 70))
    ref = np.ones((1, H, W), np.uint8); ref[:, r0:r0 + 60, c0:c0 + 70] = 2
    region = np.zeros((1, H, W), np.uint8); region[:, r0 - 10:r0 + 70, c0 - 10:c0 + 80] = 255
    return t1, t2, ref, region


def test_demo_rsss_and_wsss_run_end_to_end():
    """Demo_RSSS / Demo_WSSS loop shapes on the HIP path (tiny epochs): multi-scene region tiles,
    LR schedules, on-device evaluator over owned centres, changed/unchanged pairing."""
    from fcd_gan_pytorch_amd import datasets, demos
    rng = np.random.default_rng(5)
    scenes = []
    for hw in ((210, 230), (200, 200)):
        t1, t2, ref, region = _scene(rng, 4, *hw)
        scenes.append(datasets.RegionTileDataset(t1, t2, region=region, ref=ref, patch_size=(200, 200),
                                                 overlap_padding=(10, 10)))
    ds = datasets.MultiSceneDataset(scenes)
    logs = []
    out = demos.demo_rsss(ds, n_channels=4, epochs_g=1, epochs_adv=2, init_batch_size=2, batch_size=2, log=logs.append,
                          allow_seeded=True)
    assert len(logs) == 3 and np.isfinite(out['history']['adv']).all() and np.isfinite(out['history']['g']).all()
    cm = out['evaluator']._sync()
    assert cm.sum() == 210 * 230 + 200 * 200          # every scene pixel scored exactly once per epoch
    # WSSS: "changed" and "unchanged" tile sets of unequal size
    chg = [(torch.from_numpy(_scene(rng, 3, 176, 176)[0]), torch.from_numpy(_scene(rng, 3, 176, 176)[1])) for _ in range(3)]
    unc = []
    for _ in range(2):
        a = torch.from_numpy(rng.standard_normal((3, 176, 176)).astype(np.float32))
        unc.append((a, a + 0.05 * torch.randn_like(a)))
    out = demos.demo_wsss(chg, unc, n_channels=3, epochs_g=1, epochs_adv=1, unc_batch_size=2, batch_size=2,
                          log=logs.append, allow_seeded=True)
    assert np.isfinite(out['history']['adv']).all() and len(out['history']['g']) == 1
