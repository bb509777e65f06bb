"""Image files -> the array the reference's loaders hand to their transforms: ``imageio.imread`` of a PNG / JPEG, uint8
[H, W, C] (dmb/data/datasets/stereo/scene_flow/base.py:17-23, dmb/apis/inference.py:153-154).  Host code (decoding is not on the
hot path); PIL is the decoder in this image -- PNG decoding is lossless, so the bytes equal imageio's."""
import numpy as np


def imread(path):
    try:
        from PIL import Image
    except ImportError as e:   # loud: no silent alternative decoder
        raise ImportError("densematchingbenchmark_amd.data.imread needs PIL (Pillow) to decode %s" % path) from e
    with Image.open(path) as im:
        if im.mode in ("P", "PA", "LA", "CMYK", "YCbCr", "1"):   # palette / exotic colour models: what imageio hands back is RGB
            im = im.convert("RGB")                                 # (8-bit grey, 16-bit "I;16" -- KITTI disparities -- and "I" / "F" stay as they are)
        arr = np.array(im)          # (a writable copy: the sample dict is handed to torch.from_numpy)
    return np.ascontiguousarray(arr)      # [H, W, C], or [H, W] for a single-channel file -- as imageio returns it
