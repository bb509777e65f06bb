"""Generate tests/golden/result_reference.pkl: the on-disk result of the reference's inference API for an AcfNet-style
result dict (disps, costs, confs), assembled by the reference's OWN functions where they can be imported
(dmb.data.datasets.evaluation.stereo.eval.remove_padding) and by the statements of dmb/apis/inference.py:197-223 where the
module itself cannot (it imports mmcv): to_cpu -> per tensor F.interpolate(v / scale_factor, 1 / scale_factor, 'bilinear',
align_corners=False) -> remove_padding -> {'Result', 'OriginalData'} -> mmcv.dump (= pickle.dump(obj, file, protocol=2) for
a .pkl path).  The fixture is data (seeded tensors in, pickled dict out).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_result.py
"""
import os
import pickle
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402


def inputs():
    """Seeded inputs shared with tests/test_host_logic.py."""
    g = torch.Generator().manual_seed(77)
    Hp, Wp, D = 8, 12, 6
    result = dict(disps=[torch.rand((1, 1, Hp, Wp), generator=g) * 5 for _ in range(3)],
                  costs=[torch.randn((1, D, Hp, Wp), generator=g) for _ in range(3)],
                  confs=[torch.rand((1, 1, Hp, Wp), generator=g) for _ in range(3)])
    rs = np.random.RandomState(78)
    ori = dict(leftImage=rs.rand(6, 10, 3).astype(np.float32) * 255, rightImage=rs.rand(6, 10, 3).astype(np.float32) * 255,
               leftDisp=rs.rand(6, 10).astype(np.float32) * 5, rightDisp=None)
    return result, ori, (6, 10)


def main():
    G.import_reference()
    from dmb.data.datasets.evaluation.stereo.eval import remove_padding
    result, ori, ori_size = inputs()
    scale_factor, pad_to_shape = 1.0, (8, 12)
    for k, v in result.items():                      # inference.py:200-211
        assert isinstance(v, (tuple, list))
        for i in range(len(v)):
            vv = v[i]
            if torch.is_tensor(vv):
                vv = F.interpolate(vv * 1.0 / scale_factor, scale_factor=1.0 / scale_factor, mode='bilinear', align_corners=False)
                if pad_to_shape is not None:
                    vv = remove_padding(vv, ori_size)
                v[i] = vv
        result[k] = v
    log = {'Result': result, 'OriginalData': ori}    # inference.py:213-216
    path = os.path.join(G.OUT, "result_reference.pkl")
    with open(path, "wb") as fp:
        pickle.dump(log, fp, protocol=2)             # mmcv.dump(obj, 'x.pkl') -> PickleHandler: pickle.dump(..., protocol=2)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
