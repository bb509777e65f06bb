"""The reference's serving entry point, one stereo pair per call: drop-in for ``dmb/apis/inference.py`` (``init_model`` :60-85,
``inference_stereo`` :88-148, ``_prepare_data`` :151-188, ``_inference_single`` :191-225) -- the caller of the hot path in the
batch-1 regime the reference publishes its timings in (configs/PSMNet/ResultOfPSMNet.md:15-19).

Same arguments, same steps, same ``result.pkl``; what differs is WHERE the steps run.  Images are decoded on the host
(``data.imread``) and go to the GPU as bytes; crop / pad / normalise is one HIP launch per image (csrc/preprocess.hip); the
model is ``build_model(cfg)`` of this package (HIP backbone + cost path); its eval-mode forward is, by default, captured once per
input shape in a HIP graph and replayed (``graph="auto"``: graph_runner.wants_graph -- at batch 1 the ~190 launches of a step
cost the host more than the device, see graph_runner.py); the result is cropped with ``remove_padding`` and written by
``result_io.save_result`` in the reference's layout.  ``scale_factor`` other than 1 goes through the library's half-pixel
bilinear kernel (``ops.bilinear_scale``) and is supported where the resampled size is integral (otherwise torch's two
scale conventions differ and this raises instead of guessing)."""
import os.path as osp

import numpy as np
import torch

from .. import ops, result_io
from ..config import Config, ConfigDict
from ..data import CenterCrop, Compose, Normalize, StereoPad, ToTensor, imread
from ..disp_io import load_scene_flow_disp
from ..evaluation.stereo import remove_padding
from ..graph_runner import GraphedForward, wants_graph
from ..modeling import build_model

IMG_EXTENSIONS = ('.jpg', '.JPG', '.jpeg', '.JPEG', '.png', '.PNG', '.ppm', '.PPM', '.bmp', '.BMP')   # inference.py:22-25


def is_image_file(filename):
    return filename.endswith(IMG_EXTENSIONS)


def is_pfm_file(filename):
    return filename.endswith('.pfm')


def load_disp(item, filename, disp_div_factor=1.0):
    """inference.py:36-46: the disparity map named by ``item[filename]`` -- an image file (KITTI: 16-bit PNG, disparity * 256) or
    a SceneFlow .pfm -- as float32 divided by ``disp_div_factor``; None when the item has no such entry."""
    path = item.get(filename)
    if path is None:
        return None
    if is_image_file(path):
        raw = imread(path).squeeze()
    elif is_pfm_file(path):
        raw = load_scene_flow_disp(path)
    else:
        raise NotImplementedError
    return raw.astype(np.float32) / disp_div_factor


def init_model(config, checkpoint=None, device='cuda:0', strict=True):
    """inference.py:60-85: config file (or object) -> eval-mode model on ``device``, optionally with a checkpoint
    (``{'state_dict': ...}`` as mmcv writes it, or a bare state dict).  ``strict=True`` (default here: a key that does not match
    is an error) | False = mmcv's ``load_checkpoint`` default, which the reference calls: missing / unexpected keys are tolerated.
    A checkpoint PATH is read with ``torch.load(weights_only=False)``: mmcv checkpoints pickle a 'meta' dict next to the tensors
    (load only files you trust, as with the reference)."""
    if isinstance(config, str):
        config = Config.fromfile(config)
    elif not isinstance(config, ConfigDict):
        raise TypeError('config must be a filename or Config object, but got {}'.format(type(config)))
    model = build_model(config)
    if checkpoint is not None:
        ckpt = torch.load(checkpoint, map_location="cpu", weights_only=False) if isinstance(checkpoint, str) else checkpoint
        state = ckpt.get("state_dict", ckpt)
        state = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state.items()}   # (mmcv strips DataParallel's prefix)
        model.load_state_dict(state, strict=bool(strict))
    model.cfg = config
    model.to(device)
    model.eval()
    return model


def _resample(t, scale_factor, mult=1.0):
    """F.interpolate(t * mult, scale_factor=scale_factor, mode='bilinear', align_corners=False) where the output size is integral."""
    H, W = t.shape[-2:]
    Ho, Wo = H * scale_factor, W * scale_factor
    if abs(Ho - round(Ho)) > 1e-9 or abs(Wo - round(Wo)) > 1e-9:
        raise NotImplementedError("scale_factor %r does not map %dx%d to an integral size" % (scale_factor, H, W))
    return ops.bilinear_scale(t.contiguous(), (int(round(Ho)), int(round(Wo))), mult)


def prepare_data(item, img_transform, cfg, device):
    """inference.py:151-188: (processed sample on ``device``, original arrays as read)."""
    views = {side: imread(item['%s_image_path' % side])[:, :, :3] for side in ('left', 'right')}       # uint8 [H, W, 3]
    disps = {side: load_disp(item, '%s_disp_map_path' % side, cfg.disp_div_factor) for side in ('left', 'right')}
    oriSample = {'leftImage': views['left'].astype(np.float32), 'rightImage': views['right'].astype(np.float32),
                 'leftDisp': disps['left'], 'rightDisp': disps['right']}
    sample = {'leftImage': np.ascontiguousarray(views['left']), 'rightImage': np.ascontiguousarray(views['right']),
              'leftDisp': None if disps['left'] is None else disps['left'][np.newaxis].copy(),
              'rightDisp': None if disps['right'] is None else disps['right'][np.newaxis].copy(),
              'original_size': views['left'].shape[:2]}
    sample = img_transform(sample)
    factor = cfg.scale_factor
    for key in list(sample):
        t = sample[key]
        if not torch.is_tensor(t):
            continue
        t = t.unsqueeze(0)
        if factor != 1.0:      # inference.py:183-187: disparities scale with the image
            t = _resample(t, factor, factor if 'Disp' in key else 1.0)
        sample[key] = t.to(device)
    return sample, oriSample


def _forward(model, procData, graph):
    batch = {k: v for k, v in procData.items() if torch.is_tensor(v)}
    if graph == "auto":
        graph = wants_graph(batch)
    if not graph:
        with torch.no_grad():
            return model(batch)
    runner = getattr(model, "_dmb_graphed_forward", None)
    if runner is None:
        runner = GraphedForward(model)
        model._dmb_graphed_forward = runner
    return runner(batch)


def _inference_single(model, batchDict, img_transform, device, graph="auto", save=True):
    cfg = model.cfg.copy()
    procData, oriData = prepare_data(batchDict, img_transform, cfg, device)
    result, _ = _forward(model, procData, graph)
    assert isinstance(result, dict)
    ori_size = procData['original_size'] if cfg.pad_to_shape is not None else None
    name = batchDict['left_image_path'].split('/')[-1].split('.')[0]
    save_root = osp.join(cfg.log_dir, name)
    logData = {'Result': result_io.crop_result(result, ori_size, cfg.scale_factor), 'OriginalData': result_io.to_cpu(oriData)}
    if save:
        path = result_io.dump_log(logData, save_root)
        print('Result of {} will be saved to {}!'.format(batchDict['left_image_path'].split('/')[-1], path))
    return logData


def inference_stereo(model, batchesDict, log_dir, pad_to_shape=None, crop_shape=None, scale_factor=1.0, disp_div_factor=1.0,
                     device='cuda:0', graph="auto", save=True):
    """inference.py:88-148.  ``batchesDict``: list of dicts with left_image_path, right_image_path and optionally
    left_disp_map_path, right_disp_map_path.  Returns the list of logged dicts ({'Result', 'OriginalData'}) -- the reference
    returns None and only writes the files.  ``graph``: "auto" (default) | True | False, see the module docstring."""
    device = next(model.parameters()).device   # model device (inference.py:146)
    assert pad_to_shape is None or crop_shape is None      # inference.py:124,127: one geometric transform at most
    geometry = [StereoPad(pad_to_shape)] if pad_to_shape is not None else ([CenterCrop(crop_shape)] if crop_shape is not None else [])
    img_transform = Compose([ToTensor(device)] + geometry + [Normalize(ops.IMAGENET_MEAN, ops.IMAGENET_STD)])
    model.cfg.update({'log_dir': log_dir, 'pad_to_shape': pad_to_shape, 'crop_shape': crop_shape, 'scale_factor': scale_factor,
                      'disp_div_factor': disp_div_factor})
    out = []
    with torch.cuda.device(device):
        for batchDict in batchesDict:
            out.append(_inference_single(model, batchDict, img_transform, device, graph, save))
    return out
