"""On-disk result format of the reference's inference API, so that its viewers read our outputs unchanged.

``dmb/apis/inference.py:213-225`` writes ``result.pkl`` = pickle of
``{'Result': {'disps': [Tensor[1,1,H,W], ...], 'costs': [Tensor[1,D,H,W], ...][, 'confs': [...]]},
   'OriginalData': {'leftImage', 'rightImage', 'leftDisp', 'rightDisp'}}`` with CPU tensors cropped to the original
size (``remove_padding``), and ``tools/view_cost.py:71-84`` reads ``Result.disps[0][0, 0]``, ``Result.costs[0][0]``
and ``OriginalData.leftDisp``.  """
import os
import pickle

import torch

from .evaluation.stereo import remove_padding


def to_cpu(obj):
    if torch.is_tensor(obj):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return {k: to_cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [to_cpu(v) for v in obj]
    return obj


def crop_result(results, original_size=None, scale_factor=1.0):
    """inference.py:197-211: every tensor of every list of the result dict on the host, brought back to the input scale
    (``scale_factor``: the test-time resampling of inference.py:183-189, undone at :205-206) and cropped to ``original_size``
    (top / right padding removed)."""
    result = to_cpu(results)
    for k, v in result.items():
        if not isinstance(v, (list, tuple)):
            raise TypeError("results['%s'] must be a list of tensors (general_stereo_model.py:82-90)" % k)
        out = []
        for t in v:
            if torch.is_tensor(t):
                if scale_factor != 1.0:    # host-side post-processing of a saved result, as in the reference
                    t = torch.nn.functional.interpolate(t * 1.0 / scale_factor, scale_factor=1.0 / scale_factor, mode="bilinear",
                                                        align_corners=False)
                if original_size is not None:
                    t = remove_padding(t, original_size).contiguous()
            out.append(t)
        result[k] = out
    return result


def dump_log(log_data, save_root):
    """``mmcv.dump(logData, <save_root>/result.pkl)`` (inference.py:213-223): pickle protocol 2 = what mmcv writes for a .pkl path."""
    os.makedirs(save_root, exist_ok=True)
    path = os.path.join(save_root, "result.pkl")
    with open(path, "wb") as fp:
        pickle.dump(log_data, fp, protocol=2)
    return path


def save_result(results, original_data, save_root, original_size=None, scale_factor=1.0):
    """Write ``<save_root>/result.pkl`` in the reference's layout (inference.py:197-223); returns the path.
    Every tensor of every list (``disps``, ``costs`` and, for AcfNet, ``confs``) goes through ``crop_result``."""
    return dump_log({"Result": crop_result(results, original_size, scale_factor), "OriginalData": to_cpu(original_data)}, save_root)


def load_result(path):
    with open(path, "rb") as fp:
        return pickle.load(fp)
