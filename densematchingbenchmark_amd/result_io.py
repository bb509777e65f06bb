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


def save_result(results, original_data, save_root, original_size=None):
    """Write ``<save_root>/result.pkl`` in the reference's layout; returns the path."""
    result = to_cpu(results)
    for k, v in result.items():
        if not isinstance(v, (list, tuple)):
            raise TypeError("results['%s'] must be a list of tensors (general_stereo_model.py:82-90)" % k)
        if original_size is not None:
            result[k] = [remove_padding(t, original_size).contiguous() if torch.is_tensor(t) else t for t in v]
    log_data = {"Result": result, "OriginalData": to_cpu(original_data)}
    os.makedirs(save_root, exist_ok=True)
    path = os.path.join(save_root, "result.pkl")
    with open(path, "wb") as fp:
        pickle.dump(log_data, fp)
    return path


def load_result(path):
    with open(path, "rb") as fp:
        return pickle.load(fp)
