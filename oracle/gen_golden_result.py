"""Generate tests/golden/result_reference.pkl: the on-disk result of the reference's inference API for an AcfNet-style
result dict (disps, costs, confs), written by the reference's OWN writer -- ``dmb.apis.inference._inference_single``
(dmb/apis/inference.py:191-225: to_cpu -> per tensor the inverse test-time resampling -> remove_padding -> {'Result',
'OriginalData'} -> mkdir_or_exist -> mmcv.dump) is imported and CALLED; nothing of it is re-typed here.  What is stubbed is
only what that function does not own: the model (a callable returning the seeded result dict), ``_prepare_data`` (reads image
files: it returns the seeded arrays instead), and the third-party ``mmcv`` (absent from this image): ``mkdir_or_exist`` =
``os.makedirs(exist_ok=True)``, ``dump(obj, 'x.pkl')`` = ``pickle.dump(obj, file, protocol=2)`` -- mmcv's PickleHandler
(mmcv/fileio/handlers/pickle_handler.py: ``kwargs.setdefault('protocol', 2)``).  The fixture is data (seeded tensors in,
pickled dict out).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_result.py
"""
import importlib.util
import os
import pickle
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

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


def import_reference_inference():
    """``dmb.apis.inference`` of the reference tree with the third-party modules it imports at file scope stubbed."""
    G.import_reference()

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def mmcv_dump(obj, file, **kwargs):            # mmcv.dump for a '.pkl' path: PickleHandler.dump_to_path -> pickle.dump(protocol=2)
        kwargs.setdefault("protocol", 2)
        with open(file, "wb") as fp:
            pickle.dump(obj, fp, **kwargs)

    mm = stub("mmcv", dump=mmcv_dump, mkdir_or_exist=lambda d, mode=0o777: os.makedirs(d, mode=mode, exist_ok=True))
    mm.runner = stub("mmcv.runner", load_checkpoint=None)
    stub("imageio", imread=None)
    # file-scope imports of inference.py that _inference_single never touches (data transforms need torchvision, the flow
    # loaders need cv2): empty stand-ins; the functions this generator exercises are the reference's own
    stub("dmb.data.transforms", stereo_trans=None)
    stub("dmb.data.transforms.stereo_trans")
    stub("dmb.data.transforms.transforms", Compose=None)
    stub("dmb.data.datasets.utils", load_scene_flow_disp=None)
    pkg = types.ModuleType("dmb.apis")
    pkg.__path__ = [os.path.join(G.REF, "dmb", "apis")]
    sys.modules["dmb.apis"] = pkg                   # (the package __init__ also imports the training loop)
    spec = importlib.util.spec_from_file_location("dmb.apis.inference", os.path.join(G.REF, "dmb", "apis", "inference.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["dmb.apis.inference"] = mod
    spec.loader.exec_module(mod)
    return mod


class _Model:
    """Stand-in for a built model: ``cfg`` as _inference_single reads it, forward = the seeded result dict."""

    def __init__(self, result, log_dir):
        self.result = result
        self.cfg = G.attrdict(dict(scale_factor=1.0, pad_to_shape=(8, 12), log_dir=log_dir))

    def __call__(self, batch):
        return self.result, None


def main():
    inf = import_reference_inference()
    result, ori, ori_size = inputs()
    tmp = tempfile.mkdtemp()
    try:
        inf._prepare_data = lambda item, img_transform, cfg, device: (dict(original_size=ori_size), ori)   # (reads files)
        log = inf._inference_single(_Model(result, tmp), {"left_image_path": "somewhere/0006.png"}, None, torch.device("cpu"))
        written = os.path.join(tmp, "0006", "result.pkl")
        assert set(log) == {"Result", "OriginalData"} and os.path.exists(written)
        path = os.path.join(G.OUT, "result_reference.pkl")
        shutil.copyfile(written, path)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
