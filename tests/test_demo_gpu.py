"""One REAL stereo pair at the headline size, end to end through the reference's own data conventions and serving API:
PNG / PFM bytes (the reference's tools/demo_data pair 0, carried by tests/golden/demo_sceneflow.npz) -> data.imread / disp_io ->
ToTensor -> StereoPad -> Normalize (csrc/preprocess.hip) -> build_model(cfg) (HIP backbone + cost path, by default replayed from a
HIP graph at batch 1) -> remove_padding -> calc_error -> result.pkl; against what the reference's inference_stereo produced on the
same files (oracle/gen_golden_demo.py) and against its FP64 evaluation (the parity contract of bench.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import dmb_oracle as O
from tests._util import golden, maxdiff
from tests.test_oracle_golden import _demo_files

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# max |disparity - reference| over a whole 544x960 map: 1.6 x the reference's own self-spread at that size (tests/test_fullsize_gpu.py)
DISP_MAX_FULL = max(1e-4, 1.6 * max(float(golden("fullsize_psmnet_spread.npz")["s544_spread_full_disp%d" % k]) for k in (1, 2, 3)))
KEYS = ("epe", "1px", "2px", "3px", "5px")


def _model(dev):
    from densematchingbenchmark_amd import synthetic
    from densematchingbenchmark_amd.apis import init_model
    model = init_model(os.path.join(ROOT, "configs", "PSMNet", "scene_flow.py"), None, "cpu")
    synthetic.init_params_(model, seed=21, classif_gain=10.0)      # the generator's weights (oracle/gen_golden_demo.py)
    return model.to(dev).eval()


def test_transforms_equal_the_reference_bit_for_bit(dev, tmp_path):
    """uint8 [H, W, 3] bytes -> padded, normalised [3, 544, 960] in ONE launch, as three separate transforms, and from a float
    [3, H, W] sample: every variant equals the tensor the reference's ToTensor / StereoPad / Normalize fed its model (sampled rows
    + FP64 checksums); a centre crop equals the oracle's."""
    from densematchingbenchmark_amd import ops
    from densematchingbenchmark_amd.data import CenterCrop, Compose, Normalize, StereoPad, ToTensor, imread
    g = golden("demo_sceneflow.npz")
    paths = _demo_files(g, tmp_path)
    raw = {k: imread(paths[k + "_image_path"]) for k in ("left", "right")}
    assert raw["left"].dtype == np.uint8 and raw["left"].shape == (540, 960, 3)
    fused = Compose([ToTensor(dev), StereoPad((544, 960)), Normalize(ops.IMAGENET_MEAN, ops.IMAGENET_STD)])
    apart = [ToTensor(dev), StereoPad((544, 960)), Normalize(ops.IMAGENET_MEAN, ops.IMAGENET_STD)]
    s1 = fused(dict(leftImage=raw["left"], rightImage=raw["right"], leftDisp=None))
    s2 = dict(leftImage=raw["left"], rightImage=raw["right"])
    for t in apart:
        s2 = t(s2)
    s3 = fused(dict(leftImage=raw["left"].astype(np.float32).transpose(2, 0, 1), rightImage=raw["right"].astype(np.float32).transpose(2, 0, 1)))
    for s in (s1, s2, s3):
        li, ri = s["leftImage"], s["rightImage"]
        assert tuple(li.shape) == (3, 544, 960) and li.dtype == torch.float32
        assert np.array_equal(li[None][:, :, ::17, :].cpu().numpy(), g["left_rows"])
        assert np.array_equal(ri[None][:, :, 3::31, :].cpu().numpy(), g["right_rows"])
        assert li.double().sum().item() == g["left_sum_f64"][0] and li.double().abs().sum().item() == g["left_sum_f64"][1]
        assert ri.double().sum().item() == g["right_sum_f64"][0] and ri.double().abs().sum().item() == g["right_sum_f64"][1]
    want = O.normalize(O.center_crop(O.image_to_chw(raw["left"]), (256, 512)))
    got = Compose([ToTensor(dev), CenterCrop((256, 512)), Normalize(ops.IMAGENET_MEAN, ops.IMAGENET_STD)])(
        dict(leftImage=raw["left"], rightImage=raw["right"], leftDisp=np.zeros((1, 540, 960), np.float32)))
    assert torch.equal(got["leftImage"].cpu(), want) and tuple(got["leftDisp"].shape) == (1, 256, 512)
    from densematchingbenchmark_amd.disp_io import load_scene_flow_disp
    disp = np.ascontiguousarray(load_scene_flow_disp(paths["left_disp_map_path"]))[None]
    got = Compose([ToTensor(dev), CenterCrop((512, 896)), Normalize(ops.IMAGENET_MEAN, ops.IMAGENET_STD)])(
        dict(leftImage=raw["left"], rightImage=raw["right"], leftDisp=disp))                     # ... and the reference's own crop
    assert np.array_equal(got["leftImage"][:, ::37, :].cpu().numpy(), g["crop_left_rows"])
    assert np.array_equal(got["leftDisp"][:, ::37, :].cpu().numpy(), g["crop_disp_rows"])
    with pytest.raises(Exception):
        StereoPad((544, 960))(dict(leftImage=torch.zeros(3, 540, 960), rightImage=torch.zeros(3, 540, 960)))   # host tensors: no CPU path


def test_inference_stereo_on_the_reference_demo_pair(dev, tmp_path):
    from densematchingbenchmark_amd import result_io
    from densematchingbenchmark_amd.apis import inference_stereo
    from densematchingbenchmark_amd.evaluation import calc_error
    g = golden("demo_sceneflow.npz")
    paths = _demo_files(g, tmp_path)
    model = _model(dev)
    assert model.backbone is not None
    logs = {}
    for mode in (False, "auto"):          # eager, then the default (a HIP graph at this size)
        out_dir = tmp_path / ("log_%s" % mode)
        logged = inference_stereo(model, [paths], str(out_dir), pad_to_shape=(544, 960), graph=mode)
        assert len(logged) == 1
        saved = result_io.load_result(os.path.join(str(out_dir), "left", "result.pkl"))
        assert set(saved) == {"Result", "OriginalData"} and set(saved["Result"]) == {"disps", "costs"}
        assert set(saved["OriginalData"]) == {"leftImage", "rightImage", "leftDisp", "rightDisp"}
        assert saved["OriginalData"]["leftImage"].shape == (540, 960, 3) and saved["OriginalData"]["rightDisp"] is None
        assert float(saved["OriginalData"]["leftImage"].astype(np.float64).sum()) == g["ori_left_image_sum"][0]
        logs[mode] = saved["Result"]
    assert hasattr(model, "_dmb_graphed_forward") and len(model._dmb_graphed_forward._graphs) == 1
    for a, b in zip(logs[False]["disps"] + logs[False]["costs"], logs["auto"]["disps"] + logs["auto"]["costs"]):
        assert torch.equal(a, b)                     # the replayed graph is the eager forward, bit for bit
    res = logs["auto"]
    gt = torch.from_numpy(np.ascontiguousarray(saved["OriginalData"]["leftDisp"]))[None, None]
    assert np.array_equal(gt[0, 0, ::45].numpy(), g["ori_left_disp_rows"])
    worst = []
    for i, (d, c) in enumerate(zip(res["disps"], res["costs"])):
        assert tuple(d.shape) == tuple(g["cropped_shape"]) == (1, 1, 540, 960) and tuple(c.shape) == tuple(g["cost_shape"])
        ref, truth = torch.from_numpy(g["disp%d_s4" % i]).double(), torch.from_numpy(g["truth%d_s4" % i])
        e_hip, e_ref = (d[:, :, ::4, ::4].double() - truth).abs(), (ref - truth).abs()
        # bench.py's PARITY_CONTRACT against the reference's FP64 evaluation of the same images
        assert e_hip.max().item() <= max(1e-4, 1.25 * e_ref.max().item()), (i, e_hip.max().item(), e_ref.max().item())
        assert e_hip.mean().item() <= e_ref.mean().item(), (i, e_hip.mean().item(), e_ref.mean().item())
        assert maxdiff(d[:, :, ::4, ::4], g["disp%d_s4" % i]) <= DISP_MAX_FULL
        assert maxdiff(c[:, ::24, 5::107, :], g["cost%d_rows" % i]) <= 5e-5
        worst.append((maxdiff(d[:, :, ::4, ::4], g["disp%d_s4" % i]), e_hip.max().item(), e_ref.max().item()))
        err = calc_error(d.to(dev), gt.to(dev), 0, 192)
        want = dict(zip(KEYS, g["err%d" % i]))
        assert abs(err["epe"] - want["epe"]) <= 1e-5 * max(1.0, want["epe"]), (err, want)
        assert all(abs(err[k] - want[k]) <= 1e-3 for k in KEYS[1:]), (err, want)
    assert maxdiff(res["disps"][0], g["disp0_full"]) <= DISP_MAX_FULL
    print("demo pair: per level (max |disp - reference|, max |hip - fp64|, max |reference - fp64|) =", [tuple("%.3g" % v for v in w) for w in worst])


def test_accumulator_on_the_padded_maps_equals_calc_error_on_the_cropped_ones(dev, tmp_path):
    """The evaluation harness's form of the same numbers: padded estimates against the top-padded ground truth with
    original_size = (540, 960) -> the error dict of the cropped maps."""
    from densematchingbenchmark_amd.data import Compose, Normalize, StereoPad, ToTensor, imread
    from densematchingbenchmark_amd import ops
    from densematchingbenchmark_amd.disp_io import load_scene_flow_disp
    from densematchingbenchmark_amd.evaluation import EpeAccumulator
    g = golden("demo_sceneflow.npz")
    paths = _demo_files(g, tmp_path)
    model = _model(dev)
    sample = Compose([ToTensor(dev), StereoPad((544, 960)), Normalize(ops.IMAGENET_MEAN, ops.IMAGENET_STD)])(
        dict(leftImage=imread(paths["left_image_path"]), rightImage=imread(paths["right_image_path"])))
    with torch.no_grad():
        results, _ = model(dict(leftImage=sample["leftImage"][None], rightImage=sample["rightImage"][None]))
    gt = torch.from_numpy(np.ascontiguousarray(load_scene_flow_disp(paths["left_disp_map_path"]))).to(dev)[None, None]
    gt_padded = ops.stereo_pad_normalize(gt, (544, 960))            # ground truth padded like the images (zeros: masked by lower_bound 0)
    acc = EpeAccumulator(dev, 3, 0, 192)
    acc.update(results["disps"], gt_padded, (540, 960))
    for i, got in enumerate(acc.summary()):
        want = dict(zip(KEYS, g["err%d" % i]))
        assert abs(got["epe"] - want["epe"]) <= 1e-5 * max(1.0, want["epe"]) and all(abs(got[k] - want[k]) <= 1e-3 for k in KEYS[1:])


def test_resampling_and_crop_of_the_serving_api(dev, tmp_path):
    """inference.py's other two pre-processing branches: ``crop_shape`` (CenterCrop of images AND disparities, then Normalize) and
    ``scale_factor`` (F.interpolate(bilinear, align_corners=False) of the processed sample, disparities multiplied by the factor)
    against the oracle / torch CPU on the demo pair; a scale that does not give an integral size refuses."""
    import torch.nn.functional as F
    from densematchingbenchmark_amd import ops
    from densematchingbenchmark_amd.apis.inference import _resample, prepare_data
    from densematchingbenchmark_amd.config import ConfigDict
    from densematchingbenchmark_amd.data import CenterCrop, Compose, Normalize, ToTensor
    g = golden("demo_sceneflow.npz")
    paths = _demo_files(g, tmp_path)
    tf = Compose([ToTensor(dev), CenterCrop((512, 896)), Normalize(ops.IMAGENET_MEAN, ops.IMAGENET_STD)])
    for scale in (1.0, 0.5):
        proc, ori = prepare_data(paths, tf, ConfigDict(scale_factor=scale, disp_div_factor=1.0), dev)
        from densematchingbenchmark_amd.data import imread
        want_img = O.normalize(O.center_crop(O.image_to_chw(imread(paths["left_image_path"])), (512, 896))).unsqueeze(0)
        want_disp = O.center_crop(torch.from_numpy(np.ascontiguousarray(ori["leftDisp"]))[None], (512, 896)).unsqueeze(0)
        if scale != 1.0:
            want_img = F.interpolate(want_img, scale_factor=scale, mode="bilinear", align_corners=False)
            want_disp = F.interpolate(want_disp * scale, scale_factor=scale, mode="bilinear", align_corners=False)
        assert tuple(proc["leftImage"].shape) == tuple(want_img.shape) and proc["rightDisp"] is None
        assert maxdiff(proc["leftImage"], want_img) <= (0.0 if scale == 1.0 else 2e-6)
        assert maxdiff(proc["leftDisp"], want_disp) <= (0.0 if scale == 1.0 else 2e-5)
    with pytest.raises(NotImplementedError):
        _resample(torch.zeros(1, 1, 15, 20, device=dev), 0.3)


def test_graphed_forward_tracks_shapes_and_parameters(dev):
    """graph_runner.GraphedForward: one graph per input signature (oldest evicted), replays equal the eager forward bit for bit on
    fresh inputs, and an in-place parameter change re-captures (the captured launches hold the packed weights of their moment)."""
    from densematchingbenchmark_amd import synthetic
    from densematchingbenchmark_amd.config import Config
    from densematchingbenchmark_amd.graph_runner import GraphedForward, wants_graph
    from densematchingbenchmark_amd.modeling import build_model
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "PSMNet", "baseline_cfg0_256x512_d64.py"))
    model = build_model(cfg).eval()                      # backbone included: two view streams inside the capture
    synthetic.init_params_(model, seed=3, classif_gain=10.0)
    model = model.to(dev)
    runner = GraphedForward(model, max_graphs=2)

    def images(seed, h, w):
        gen = torch.Generator().manual_seed(seed)
        return dict(leftImage=torch.randn((1, 3, h, w), generator=gen).to(dev), rightImage=torch.randn((1, 3, h, w), generator=gen).to(dev))

    def same(batch):
        got = [t.clone() for t in runner(batch)[0]["disps"]]
        with torch.no_grad():
            want = model(batch)[0]["disps"]
        return all(torch.equal(a, b) for a, b in zip(got, want))

    # (the PSMNet backbone pools 64 x 64 windows at quarter resolution: 256 rows / columns is its smallest input)
    assert wants_graph(images(0, 256, 512)) and not wants_graph(dict(leftImage=torch.zeros(1, 3, 8, 8)))
    assert same(images(1, 256, 512)) and same(images(2, 256, 512)) and len(runner._graphs) == 1
    assert same(images(3, 256, 256)) and len(runner._graphs) == 2
    assert same(images(4, 320, 256)) and len(runner._graphs) == 2            # the 256x512 graph was evicted
    with torch.no_grad():
        model.cost_processor.aggregator.classif3[1].weight.mul_(1.5)         # in place: version bump, same storage
    before = runner._graphs
    assert same(images(5, 320, 256))
    assert len(runner._graphs) == 1 and runner._graphs is before             # reset + one fresh capture
    model.train()
    with pytest.raises(RuntimeError):
        GraphedForward(model)(images(6, 256, 256))
