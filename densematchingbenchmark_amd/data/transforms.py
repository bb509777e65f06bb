"""The reference's sample transforms in front of the model, on the device: drop-in names for
dmb/data/transforms/stereo_trans.py (ToTensor :9-18, CenterCrop :20-44, Normalize :78-90, StereoPad :92-119) and
transforms.py's Compose, in the evaluation order of dmb/data/datasets/stereo/builder.py:22-28
(ToTensor -> StereoPad -> Normalize: padding BEFORE normalisation, so padded pixels hold -mean/std).

A sample is the reference's dict ('leftImage', 'rightImage' [C, H, W], optional 'leftDisp' / 'rightDisp' [1, H, W],
'original_size').  ``ToTensor(device)`` moves the arrays to the GPU once -- images may stay the decoder's uint8 [H, W, C] bytes
(``imread``), a quarter of the PCIe traffic; everything after it is csrc/preprocess.hip (``ops.stereo_pad_normalize``): one launch
per image for crop / pad / normalise, and ``Compose`` runs a StereoPad followed by a Normalize as ONE launch.  Arithmetic equal to
the reference's, bit for bit (FP32 subtract, correctly rounded FP32 divide).  There is no CPU path: host tensors raise."""
import numbers

import numpy as np
import torch

from .. import ops

IMAGE_KEYS = ("leftImage", "rightImage")


def _as_batch(img):
    """[C, H, W] float / [H, W, C] uint8 -> the 4-D tensor the kernel takes (a view) and whether it was unbatched."""
    if img.dim() == 3:
        return img.unsqueeze(0), True
    return img, False


def _hw(img):
    return (img.shape[-3], img.shape[-2]) if img.dtype == torch.uint8 else (img.shape[-2], img.shape[-1])


class ToTensor(object):
    """numpy arrays -> tensors on ``device`` (stereo_trans.py:9-18 converts on the host; here the sample goes to the GPU once).
    uint8 images in the decoder's [H, W, C] layout stay bytes: the first kernel that touches them converts."""

    def __init__(self, device="cuda"):
        self.device = torch.device(device)

    def __call__(self, sample):
        for k in sample.keys():
            v = sample[k]
            if v is not None and isinstance(v, np.ndarray):
                sample[k] = torch.from_numpy(np.ascontiguousarray(v)).to(self.device, non_blocking=True)
            elif torch.is_tensor(v):
                sample[k] = v.to(self.device, non_blocking=True)
        return sample


class _Geometry(object):
    def __init__(self, size):
        self.size = (int(size), int(size)) if isinstance(size, numbers.Number) else tuple(int(v) for v in size)


class StereoPad(_Geometry):
    """stereo_trans.py:92-119: zeros on the TOP and on the RIGHT of both images up to ``size``; disparities are not padded."""

    def __call__(self, sample, normalize=None):
        th, tw = self.size
        for k in IMAGE_KEYS:
            x, single = _as_batch(sample[k])
            h, w = _hw(x)
            if (h, w) == (th, tw) and normalize is None and x.dtype == torch.float32:
                continue
            mean, std = (normalize.mean, normalize.std) if normalize is not None else (None, None)
            y = ops.stereo_pad_normalize(x, (th, tw), mean, std)
            sample[k] = y[0] if single else y
        return sample


class CenterCrop(_Geometry):
    """stereo_trans.py:20-44: the central ``size`` window of every array of the sample (images and disparities)."""

    def __call__(self, sample):
        th, tw = self.size
        h, w = _hw(_as_batch(sample["leftImage"])[0])
        if (h, w) == (th, tw):
            return sample
        y1, x1 = (h - th) // 2, (w - tw) // 2
        for k, v in sample.items():
            if v is None or not torch.is_tensor(v):
                continue
            x, single = _as_batch(v)
            y = ops.stereo_pad_normalize(x, None, None, None, window=(y1, x1, th, tw), channels=(x.shape[-1] if x.dtype == torch.uint8 else None))
            sample[k] = y[0] if single else y
        return sample


class Normalize(object):
    """stereo_trans.py:78-90: (x - mean[c]) / std[c] on both images."""

    def __init__(self, mean, std):
        self.mean, self.std = [float(v) for v in mean], [float(v) for v in std]

    def __call__(self, sample):
        for k in IMAGE_KEYS:
            x, single = _as_batch(sample[k])
            y = ops.stereo_pad_normalize(x, None, self.mean, self.std)
            sample[k] = y[0] if single else y
        return sample


class Compose(object):
    """transforms.py's Compose; a StereoPad directly followed by a Normalize runs as one launch per image."""

    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, sample):
        i, ts = 0, self.transforms
        while i < len(ts):
            if isinstance(ts[i], StereoPad) and i + 1 < len(ts) and isinstance(ts[i + 1], Normalize):
                sample = ts[i](sample, normalize=ts[i + 1])
                i += 2
            else:
                sample = ts[i](sample)
                i += 1
        return sample


def build_transforms(cfg, type, is_train=False, device="cuda"):
    """dmb/data/datasets/stereo/builder.py:8-31, evaluation side (the training-side RandomCrop belongs to the data loader,
    which is outside the path)."""
    if is_train:
        raise NotImplementedError("training-side sample transforms (RandomCrop) live in the data loader, outside the HIP path")
    node = cfg.data[type]
    return Compose([ToTensor(device), StereoPad(node.input_shape), Normalize(node.mean, node.std)])
