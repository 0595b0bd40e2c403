"""Host-side data formats of the reference: what `lib/data/clevr.py:23-50`,
`lib/data/dsprite.py:15-28` and `lib/data/build.py:19-31` hand to the model and to the ARI evaluator.

The reference builds these from torchvision transforms over PIL images (torchvision / skimage are not needed here: the
arithmetic is PIL's, which this module calls directly):

    CLEVR image : uint8 (H, W, >=3)  -> ToPILImage -> CenterCrop(192) -> Resize(128) [PIL bilinear, antialiased]
                  -> ToTensor                                       => float32 (3, 128, 128) in [0, 1]
    CLEVR masks : colour-coded (H, W, 3) -> one boolean mask per colour except the (64, 64, 64) background
                  -> CenterCrop(192) -> Resize(128, NEAREST)         => float32 (K_gt, 128, 128) in {0, 1}
    dSprites    : uint8 (64, 64, 3) -> ToTensor; masks from .npy     => float32 (3, 64, 64), (K_gt, 64, 64)
    batches     : images stacked, masks kept as a per-image list (their K_gt differs)  (`collate_fn`)

Mask order: the reference iterates a Python `set` of colours (arbitrary order); masks here are sorted by colour.  ARI is
invariant to the order.
"""
import os

import numpy as np
import torch

CLEVR_BACKGROUND = (64, 64, 64)


def _pil():
    from PIL import Image
    return Image


def center_crop_box(h: int, w: int, size: int):
    """torchvision.transforms.functional.center_crop: top/left rounded like torchvision does."""
    top = int(round((h - size) / 2.0))
    left = int(round((w - size) / 2.0))
    return left, top, left + size, top + size


def clevr_image(img: np.ndarray, crop: int = 192, size: int = 128) -> torch.Tensor:
    """clevr.py:25-31."""
    Image = _pil()
    img = np.ascontiguousarray(img[:, :, :3]).astype(np.uint8)
    pil = Image.fromarray(img).crop(center_crop_box(img.shape[0], img.shape[1], crop))
    pil = pil.resize((size, size), Image.BILINEAR)
    return torch.from_numpy(np.asarray(pil, dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)


def clevr_separate_masks(mask_img: np.ndarray) -> np.ndarray:
    """clevr.py:55-84 (`sep`): one boolean mask per colour, background colour removed.  (K_gt, H, W)"""
    m = np.ascontiguousarray(mask_img[:, :, :3])
    colours = np.unique(m.reshape(-1, 3), axis=0)
    out = [np.all(m == c, axis=2) for c in colours if tuple(int(v) for v in c) != CLEVR_BACKGROUND]
    return np.stack(out, axis=0) if out else np.zeros((0,) + m.shape[:2], dtype=bool)


def clevr_masks(mask_img: np.ndarray, crop: int = 192, size: int = 128) -> torch.Tensor:
    """clevr.py:33-47: separated masks, centre-cropped, nearest-resized; float32 (K_gt, size, size)."""
    Image = _pil()
    masks = clevr_separate_masks(mask_img)
    box = center_crop_box(mask_img.shape[0], mask_img.shape[1], crop)
    out = [np.asarray(Image.fromarray(x.astype(np.uint8)).crop(box).resize((size, size), Image.NEAREST)) for x in masks]
    arr = np.stack(out, axis=0) if out else np.zeros((0, size, size), dtype=np.uint8)
    return torch.from_numpy(arr.astype(np.float32))


def dsprites_item(img: np.ndarray, mask: np.ndarray):
    """dsprite.py:15-28: ToTensor on the image, masks as stored."""
    x = torch.from_numpy(np.array(img, dtype=np.uint8)).permute(2, 0, 1).float().div_(255.0)
    return x, torch.from_numpy(mask.astype(np.float32))


def collate_fn(batch):
    """build.py:19-31: images stacked to (B, 3, S, S); masks stay a tuple (K_gt differs per image)."""
    data, mask = zip(*batch)
    return torch.stack(data, dim=0), mask


class CLEVR(torch.utils.data.Dataset):
    """`root/images/*.png` (+ optional `root/masks/<same name>`), clevr.py:10-53."""

    def __init__(self, root, mode='train'):
        assert os.path.exists(root), 'Path {} does not exist'.format(root)
        self.root = root
        self.img_paths = sorted(f.path for f in os.scandir(os.path.join(root, 'images')))

    def __len__(self):
        return len(self.img_paths)

    def __getitem__(self, index):
        Image = _pil()
        path = self.img_paths[index]
        img = clevr_image(np.asarray(Image.open(path).convert('RGB')))
        mask_path = os.path.join(self.root, 'masks', os.path.split(path)[-1])
        mask = clevr_masks(np.asarray(Image.open(mask_path).convert('RGB'))) if os.path.exists(mask_path) else None
        return img, mask


class MultiDSprites(torch.utils.data.Dataset):
    """`root/images/{i}.png`, `root/masks/{i}.npy`, dsprite.py:11-32."""

    def __init__(self, root, mode='train', length=60000):
        self.root, self.length = root, length

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        Image = _pil()
        img = np.asarray(Image.open(os.path.join(self.root, 'images/{}.png'.format(index))).convert('RGB'))
        mask = np.load(os.path.join(self.root, 'masks/{}.npy'.format(index)))
        return dsprites_item(img, mask)


def make_dataloader(dataset, batch_size, shuffle, num_workers=0, rank=0, world_size=1):
    """build.py:6-17 with one process per GPU: each rank iterates its own shard of the index set."""
    sampler = None
    if world_size > 1:
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=world_size, rank=rank, shuffle=shuffle)
        shuffle = False
    return torch.utils.data.DataLoader(dataset, batch_size=batch_size, collate_fn=collate_fn, shuffle=shuffle,
                                       sampler=sampler, num_workers=num_workers)
