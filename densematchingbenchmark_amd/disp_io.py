"""Ground-truth disparity files of the SceneFlow datasets (PFM), so that the evaluation harness can be fed what the reference's
data pipeline feeds its own: drop-in for ``dmb/data/datasets/utils/load_disp.py:5-68`` (``load_pfm``,
``load_scene_flow_disp``).  (backlog row "on-disk formats".)

Format (as the reference parses it): line 1 ``PF`` (3 channels) or ``Pf`` (1 channel); line 2 ``<width> <height>``; line 3 a
scale whose SIGN is the byte order (negative = little-endian) and whose magnitude is returned; then height * width
(* 3) float32 values, BOTTOM row first -- the loader flips to top-down.  Host code only: no GPU work, no torch."""
import re

import numpy as np


def load_pfm(file_path):
    """-> (data [H, W] or [H, W, 3] float32, top row first, in the file's byte order as the reference returns it; scale)."""
    with open(file_path, "rb") as fp:
        header = fp.readline().decode("ISO-8859-1").rstrip()
        if header == "PF":
            color = True
        elif header == "Pf":
            color = False
        else:
            raise Exception("Not a PFM file.")                      # (the reference's exception type and text)
        dim_match = re.match(r"^(\d+)\s(\d+)\s$", fp.readline().decode("ISO-8859-1"))
        if not dim_match:
            raise Exception("Malformed PFM header.")
        width, height = map(int, dim_match.groups())
        scale = float(fp.readline().decode("ISO-8859-1").rstrip())
        if scale < 0:      # little-endian
            endian, scale = "<", -scale
        else:
            endian = ">"
        data = np.frombuffer(fp.read(), dtype=endian + "f")
    shape = (height, width, 3) if color else (height, width)
    return np.flipud(np.reshape(data, shape)), scale


def load_scene_flow_disp(img_path):
    """load_disp.py:57-68: the disparity map of a SceneFlow ``.pfm`` file, [H, W]."""
    assert img_path.endswith(".pfm"), "scene flow disparity image must end with .pfm" "but got {}".format(img_path)
    disp_img, __ = load_pfm(img_path)
    return disp_img


def write_pfm(file_path, data, scale=1.0, little_endian=True):
    """The inverse of load_pfm (tests and synthetic ground truth): ``data`` [H, W] or [H, W, 3], top row first."""
    data = np.asarray(data, dtype=np.float32)
    if data.ndim == 3 and data.shape[2] == 3:
        tag = "PF"
    elif data.ndim == 2:
        tag = "Pf"
    else:
        raise ValueError("PFM holds [H, W] or [H, W, 3] arrays, got %s" % (data.shape,))
    with open(file_path, "wb") as fp:
        fp.write(("%s\n%d %d\n%s\n" % (tag, data.shape[1], data.shape[0], repr(-abs(scale) if little_endian else abs(scale)))).encode("ISO-8859-1"))
        fp.write(np.flipud(data).astype(("<" if little_endian else ">") + "f4").tobytes())
