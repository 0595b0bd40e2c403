"""Data formats of the reference (lib/data/clevr.py:23-50, dsprite.py:15-28, build.py:19-31) on synthetic files."""
import numpy as np
import pytest
import torch

from iodine_amd import data as D
from iodine_amd.ari import compute_ari

PIL = pytest.importorskip('PIL')


def _scene(h=320, w=480, seed=0):
    """CLEVR-sized synthetic frame + colour-coded mask image (background (64, 64, 64))."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)               # RGBA like the CLEVR PNGs
    mask = np.full((h, w, 3), 64, dtype=np.uint8)
    cols = [(255, 0, 0), (0, 255, 0), (0, 0, 255), (10, 200, 30)]
    for i, c in enumerate(cols):
        y, x = 80 + 40 * i, 160 + 50 * i
        mask[y:y + 60, x:x + 70] = c
    return img, mask, cols


def test_center_crop_box_matches_torchvision_rounding():
    assert D.center_crop_box(320, 480, 192) == (144, 64, 336, 256)
    assert D.center_crop_box(193, 193, 192) == (0, 0, 192, 192)              # round(0.5) == 0 (banker's), as torchvision


def test_clevr_image_shape_range_and_constant():
    img, _, _ = _scene()
    x = D.clevr_image(img)
    assert x.shape == (3, 128, 128) and x.dtype == torch.float32
    assert 0.0 <= float(x.min()) and float(x.max()) <= 1.0
    flat = np.full((320, 480, 3), 200, dtype=np.uint8)
    assert torch.equal(D.clevr_image(flat), torch.full((3, 128, 128), 200 / 255.0))
    # only the 192x192 centre contributes
    a = img.copy(); a[:64] = 0; a[256:] = 0; a[:, :144] = 0; a[:, 336:] = 0
    assert torch.equal(D.clevr_image(a), x)


def test_clevr_masks_separation_crop_and_nearest():
    _, mask, cols = _scene()
    m = D.clevr_masks(mask)
    assert m.shape == (len(cols), 128, 128) and set(np.unique(m.numpy())) <= {0.0, 1.0}
    sep = D.clevr_separate_masks(mask)
    assert sep.shape[0] == len(cols) and not sep.sum(0).max() > 1          # disjoint, background dropped
    # PIL NEAREST 192 -> 128 samples source index floor((i + 0.5) * 1.5) of the cropped mask
    idx = np.floor((np.arange(128) + 0.5) * 1.5).astype(int)
    crop = sep[:, 64:256, 144:336]
    assert np.array_equal(m.numpy(), crop[:, idx][:, :, idx].astype(np.float32))
    # a perfect prediction scores ARI 1 against these masks (contingency table gt x pred over foreground pixels)
    onehot = m.numpy().reshape(len(cols), -1)
    assert abs(compute_ari(onehot @ onehot.T) - 1.0) < 1e-12


def test_datasets_and_collate(tmp_path):
    from PIL import Image
    root = tmp_path / 'CLEVR'
    (root / 'images').mkdir(parents=True); (root / 'masks').mkdir()
    for i in range(3):
        img, mask, _ = _scene(seed=i)
        Image.fromarray(img).save(root / 'images' / f'{i}.png')
        if i != 1:
            Image.fromarray(mask).save(root / 'masks' / f'{i}.png')
    ds = D.CLEVR(str(root))
    assert len(ds) == 3
    x0, m0 = ds[0]
    assert x0.shape == (3, 128, 128) and m0.shape == (4, 128, 128)
    assert ds[1][1] is None                                                  # no mask file -> None (clevr.py:41)
    xs, ms = next(iter(D.make_dataloader(ds, batch_size=3, shuffle=False)))
    assert xs.shape == (3, 3, 128, 128) and isinstance(ms, tuple) and ms[1] is None
    droot = tmp_path / 'DS'
    (droot / 'images').mkdir(parents=True); (droot / 'masks').mkdir()
    rng = np.random.default_rng(0)
    Image.fromarray(rng.integers(0, 256, size=(64, 64, 3), dtype=np.uint8)).save(droot / 'images' / '0.png')
    np.save(droot / 'masks' / '0.npy', rng.integers(0, 2, size=(3, 64, 64)))
    x, m = D.MultiDSprites(str(droot), length=1)[0]
    assert x.shape == (3, 64, 64) and m.shape == (3, 64, 64) and m.dtype == torch.float32
