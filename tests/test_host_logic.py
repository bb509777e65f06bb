"""CPU-side tests: C-ABI exports, config loader, registries, state-dict interop, loud failure without a GPU."""
import ctypes
import os

import numpy as np
import pytest
import torch

from densematchingbenchmark_amd import _lib, ops
from densematchingbenchmark_amd.config import Config
from tests._util import golden

PSM_CFG = dict(
    model=dict(
        meta_architecture="GeneralizedStereoModel", max_disp=192, batch_norm=True,
        cost_processor=dict(type='Concatenation',
                            cost_computation=dict(type="default", max_disp=48, start_disp=0, dilation=1),
                            cost_aggregator=dict(type="PSMNet", max_disp=192, in_planes=64)),
        disp_predictor=dict(type='FASTER', max_disp=192, start_disp=0, dilation=1, alpha=1.0, normalize=True),
    ))


def _cfg(**over):
    cfg = Config(PSM_CFG)
    for k, v in over.items():
        node = cfg
        parts = k.split("__")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return cfg


def test_library_exports_every_header_symbol():
    lib = _lib.load()
    syms = _lib.header_symbols()
    assert len(syms) >= 23
    for s in syms:
        assert hasattr(lib, s), "libdmb_hip.so does not export %s" % s
    assert set(syms) == set(_lib.SIGNATURES), "ctypes table and include/dmb_hip.h disagree"
    assert lib.dmb_abi_version() == 8 == _lib.ABI_VERSION
    assert lib.dmb_conv3d_packed_floats(32, 64) == 32 * 64 * 27


def test_release_library_exports_exactly_the_header():
    """The dynamic symbol table of libdmb_hip.so against include/dmb_hip.h: every declared entry point and NOTHING else named
    dmb_* -- in particular no dmb_dev_set_option and no option table (development switches exist only in the -DDMB_DEV build,
    lib/libdmb_hip_dev.so, which nothing here loads)."""
    import shutil
    import subprocess
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = {line.split()[-1] for line in out.splitlines() if line.strip()}
    exported = {n for n in names if n.startswith("dmb_")}
    assert exported == set(_lib.header_symbols()), (sorted(exported - set(_lib.header_symbols())), sorted(set(_lib.header_symbols()) - exported))
    assert not any("dev_opts" in n or "dev_set_option" in n for n in names)
    assert not _lib.DEV_BUILD and _lib.LIB_PATH.endswith("libdmb_hip.so")


def test_argument_validation_without_gpu():
    """Entry points validate arguments before touching the device, so this runs on a CPU-only box."""
    lib = _lib.load()
    idx = ops.host_ints([0])
    assert lib.dmb_cat_fms_f32(None, None, None, 1, 1, 1, 1, 1, idx, None) == 100001
    assert b"volume" in lib.dmb_last_error()
    assert lib.dmb_conv3d_k3_f32(None, None, None, None, None, None, 1, 32, 32, 4, 4, 4, 1, 0, None) == 100001
    assert lib.dmb_soft_argmin_f32(None, None, 1, 4, 2, 2, 1.0, 1, ops.host_floats([0, 1, 2, 3]), None) == 100001


def test_no_cpu_fallback():
    x = torch.zeros(1, 4, 2, 2)
    with pytest.raises(_lib.DmbLibraryError, match="no CPU fallback"):
        ops.soft_argmin(x, [0.0, 1.0, 2.0, 3.0])
    with pytest.raises(_lib.DmbLibraryError):
        ops.cat_fms(torch.zeros(1, 2, 3, 4), torch.zeros(1, 2, 3, 4), [0, 1])


def test_disp_lists_match_reference_convention():
    assert ops.disp_index_list(6, 0, 2) == [0, 2, 5]          # truncation of an FP32 linspace
    assert ops.disp_index_list(5, -2, 2) == [-2, 0, 2]
    assert ops.disp_sample_values(9, -4, 2) == [-4.0, -2.0, 0.0, 2.0, 4.0]


def test_config_loader_reads_reference_style_files(tmp_path):
    f = tmp_path / "cfg.py"
    f.write_text("import os.path as osp\nmax_disp = 64\nmodel = dict(max_disp=max_disp, batch_norm=True,\n"
                 "  cost_processor=dict(type='Concatenation', cost_computation=dict(type='default', max_disp=int(max_disp // 4))))\n")
    cfg = Config.fromfile(str(f))
    assert cfg.model.cost_processor.cost_computation.max_disp == 16
    c = cfg.model.cost_processor.cost_computation.copy()
    assert c.pop('type') == 'default' and 'type' in cfg.model.cost_processor.cost_computation
    assert cfg.model.get('cmn') is None and 'osp' not in cfg


def test_registries_and_builders():
    from densematchingbenchmark_amd.modeling.stereo.cost_processors import PROCESSORS, build_cost_processor
    from densematchingbenchmark_amd.modeling.stereo.cost_processors.aggregators import AGGREGATORS
    from densematchingbenchmark_amd.modeling.stereo.cost_processors.utils.cat_fms import CAT_FUNCS
    from densematchingbenchmark_amd.modeling.stereo.cost_processors.utils.dif_fms import DIF_FUNCS
    from densematchingbenchmark_amd.modeling.stereo.cost_processors.utils.gwc_fms import COR_FUNCS
    from densematchingbenchmark_amd.modeling.stereo.disp_predictors import PREDICTORS, build_disp_predictor
    assert set(PROCESSORS) == {'Difference', 'Concatenation', 'Correlation'}
    assert {'PSMNet', 'AcfNet', 'StereoNet'} <= set(AGGREGATORS)
    assert set(CAT_FUNCS) == {'default', 'fast_mode'} and set(DIF_FUNCS) == {'default', 'fast_mode'}
    assert set(COR_FUNCS) == {'default', 'gwc', 'gwc_cat'} and set(PREDICTORS) == {'DEFAULT', 'FASTER', 'LOCAL'}
    # the reference's own entry keeps the reference's semantics (correlation1d_cost.py:29-31); the group-wise volumes have their own keys
    assert COR_FUNCS['default'].__name__ == 'correlation1d_cost' and COR_FUNCS['gwc'].__name__ == 'gwc_fms'
    cp = build_cost_processor(_cfg())
    assert cp.default_args == dict(max_disp=48, start_disp=0, dilation=1)
    assert type(cp.aggregator).__name__ == 'PSMAggregator' and cp.aggregator.max_disp == 192
    with pytest.raises(AssertionError):
        build_cost_processor(_cfg(model__cost_processor__type='Nope'))
    with pytest.raises(NotImplementedError):
        build_cost_processor(_cfg(model__cost_processor__type='AnyNet'))
    dp = build_disp_predictor(_cfg(model__disp_predictor__type='LOCAL', model__disp_predictor__radius=3))
    assert dp.name == 'LocalSoftArgmin' and dp.radius == 3
    with pytest.raises(ValueError, match="expected 4D input"):
        build_disp_predictor(_cfg())(torch.zeros(2, 3, 4))


@pytest.mark.parametrize("tag,rel", [("psmnet", "PSMNet/scene_flow.py"), ("acfnet", "AcfNet/scene_flow_adaptive.py"),
                                     ("stereonet", "StereoNet/scene_flow_8x_2stage.py")])
def test_state_dict_keys_match_reference(tag, rel):
    """Checkpoint interop: same parameter/buffer names and shapes as the reference's modules."""
    from densematchingbenchmark_amd.modeling import build_model
    want = set(str(s) for s in golden("state_dict_keys.npz")[tag])
    cfg_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", rel)
    cfg = Config.fromfile(cfg_path)
    model = build_model(cfg, backbone=None)
    got = set("%s %s" % (k, tuple(v.shape)) for k, v in model.state_dict().items())
    assert got == want, (sorted(got - want)[:5], sorted(want - got)[:5])
    n = {"psmnet": 154, "acfnet": 185, "stereonet": 31}[tag]
    assert len(got) == n


def test_bn_fold_matches_batch_norm():
    from densematchingbenchmark_amd.modeling.stereo.layers.basic_layers import fold_batch_norm
    bn = torch.nn.BatchNorm3d(8).eval()
    g = torch.Generator().manual_seed(0)
    bn.weight.data, bn.bias.data = torch.rand(8, generator=g) + 0.5, torch.rand(8, generator=g) - 0.5
    bn.running_mean, bn.running_var = torch.rand(8, generator=g) - 0.5, torch.rand(8, generator=g) + 0.5
    bias = torch.rand(8, generator=g)
    x = torch.randn(2, 8, 3, 4, 5, generator=g)
    scale, shift = fold_batch_norm(bn, bias, 8, x.device)
    ref = bn(x + bias.view(1, -1, 1, 1, 1))
    assert (x * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1) - ref).abs().max().item() <= 1e-6


def test_modules_refuse_cpu_tensors():
    from densematchingbenchmark_amd.modeling import build_model
    m = build_model(_cfg(), backbone=None)
    feats = dict(leftFeature=torch.zeros(1, 32, 16, 16), rightFeature=torch.zeros(1, 32, 16, 16))
    with pytest.raises(_lib.DmbLibraryError):    # training mode runs the same HIP kernels under autograd: no CPU path either
        m.train()(feats)
    with pytest.raises(_lib.DmbLibraryError):
        m.eval()(feats)


def test_remove_padding_views():
    from densematchingbenchmark_amd.evaluation import remove_padding
    x = torch.arange(2 * 1 * 6 * 8, dtype=torch.float32).view(2, 1, 6, 8)
    y = remove_padding(x, (4, 5))
    assert y.shape == (2, 1, 4, 5) and torch.equal(y, x[:, :, 2:, :5])
    assert remove_padding(x, (9, 5)).shape == x.shape            # negative pad_top: untouched (eval.py:26-29)
    assert remove_padding({"a": [x]}, (4, 5))["a"][0].shape == (2, 1, 4, 5)


def test_result_pkl_layout_matches_reference_reader(tmp_path):
    """tools/view_cost.py:71-84 access pattern on a file written by save_result."""
    import pickle
    from densematchingbenchmark_amd.result_io import save_result
    res = dict(disps=[torch.rand(1, 1, 8, 12)], costs=[torch.rand(1, 6, 8, 12)])
    ori = dict(leftImage=torch.zeros(3, 6, 10), rightImage=torch.zeros(3, 6, 10), leftDisp=torch.rand(6, 10), rightDisp=None)
    path = save_result(res, ori, str(tmp_path / "pair0"), original_size=(6, 10))
    with open(path, "rb") as fp:
        r = pickle.load(fp)
    est = r['Result']['disps'][0][0, 0, ].cpu().numpy()
    vol = r['Result']['costs'][0][0].cpu().numpy()
    assert est.shape == (6, 10) and vol.shape == (6, 6, 10) and r['OriginalData']['leftDisp'].shape == (6, 10)
    assert np.array_equal(est, res['disps'][0][0, 0, 2:, :10].numpy())   # top/right padding removed (eval.py:24-29)


def test_result_pkl_equals_the_reference_writers_file(tmp_path):
    """save_result on an AcfNet-style result (disps, costs AND confs) against the file the reference's writer produces for
    the same inputs (tests/golden/result_reference.pkl: dmb/apis/inference.py:197-223 run by oracle/gen_golden_result.py),
    then read back with the literal access pattern of tools/view_cost.py:71-101,132-150."""
    import pickle
    from densematchingbenchmark_amd.result_io import load_result, save_result
    from oracle.gen_golden_result import inputs
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "result_reference.pkl"), "rb") as fp:
        ref = pickle.load(fp)
    result, ori, ori_size = inputs()
    mine = load_result(save_result(result, ori, str(tmp_path / "0006"), original_size=ori_size, scale_factor=1.0))
    assert set(mine) == set(ref) == {"Result", "OriginalData"}
    assert set(mine["Result"]) == set(ref["Result"]) == {"disps", "costs", "confs"}
    for k in ref["Result"]:
        assert isinstance(mine["Result"][k], list) and len(mine["Result"][k]) == len(ref["Result"][k]) == 3
        for a, b in zip(mine["Result"][k], ref["Result"][k]):
            assert a.device.type == "cpu" and a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b)
    assert set(mine["OriginalData"]) == set(ref["OriginalData"])
    for k, b in ref["OriginalData"].items():
        a = mine["OriginalData"][k]
        assert (a is None and b is None) or (type(a) is type(b) and np.array_equal(a, b))
    # tools/view_cost.py:71-84
    ori_data, net_result = mine['OriginalData'], mine['Result']
    leftImage, gtDisp = ori_data['leftImage'], ori_data['leftDisp']
    estDisp = net_result['disps'][0][0, 0, ].cpu().numpy()
    costVolume = net_result['costs'][0][0].cpu().numpy()
    err_map = np.abs(gtDisp - estDisp)
    assert err_map.shape == (6, 10) == leftImage.shape[:2] and costVolume.shape == (6, 6, 10)
    # tools/view_cost.py:132-150: the distribution at a pixel
    prob = torch.softmax(torch.from_numpy(costVolume), dim=0)
    h, w = 3, 4
    assert abs(float(prob[:, h, w].sum()) - 1.0) < 1e-6 and np.isfinite(abs(gtDisp[h, w] - estDisp[h, w]))
    assert net_result['confs'][2].shape == (1, 1, 6, 10)


def test_registry_instantiate():
    from densematchingbenchmark_amd.modeling.registry import UnknownType, instantiate

    class A:
        def __init__(self, x, batch_norm=False):
            self.x, self.batch_norm = x, batch_norm

    table = dict(a=A)
    obj = instantiate(table, dict(type="a", x=3), "thing", batch_norm=True)
    assert isinstance(obj, A) and obj.x == 3 and obj.batch_norm is True
    assert instantiate(table, dict(x=1), "thing", default_type="a").x == 1
    with pytest.raises(NotImplementedError):
        instantiate(table, dict(type="AnyNet"), "thing", off_path=("AnyNet",))
    with pytest.raises(UnknownType):
        instantiate(table, dict(type="zzz"), "thing")
    node = dict(type="a", x=5)
    instantiate(table, node, "thing")
    assert node == dict(type="a", x=5)          # the config node is not consumed


def test_opt_in_conv3d_mode_is_explicit():
    """The split-bf16 convolution is never selected implicitly: the default mode is 'exact' and only the two documented
    values are accepted."""
    assert ops.conv3d_mode() == "exact"
    with pytest.raises(ValueError):
        ops.set_conv3d_mode("fast")
    ops.set_conv3d_mode("bf16x6")
    try:
        assert ops.conv3d_mode() == "bf16x6"
    finally:
        ops.set_conv3d_mode("exact")
    assert ops.conv3d_mode() == "exact"


def _keys(tag):
    return set(str(s) for s in golden("state_dict_keys.npz")[tag])


@pytest.mark.parametrize("rel,want", [
    ("PSMNet/scene_flow.py", ("psmnet", "psmnet_backbone")),
    ("AcfNet/scene_flow_uniform.py", ("acfnet-cmn", "psmnet_backbone")),
    ("StereoNet/scene_flow_8x_refined.py", ("stereonet", "stereonet_backbone", "stereonet_refinement")),
    ("GCNet/scene_flow.py", ("gcnet",)),
])
def test_whole_model_configs_match_reference_keys(rel, want):
    """build_model(cfg, backbone="hip") on the whole-model configs: the union of the reference's key lists (checkpoint interop
    for the training / end-to-end entry points; constructing the modules needs no GPU)."""
    from densematchingbenchmark_amd.modeling import build_model
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", rel))
    model = build_model(cfg, backbone="hip")
    got = set("%s %s" % (k, tuple(v.shape)) for k, v in model.state_dict().items())
    expect = set()
    for tag in want:
        if tag == "acfnet-cmn":     # the fixed-variance config has no confidence network
            expect |= set(k for k in _keys("acfnet") if not k.startswith("cmn."))
        else:
            expect |= _keys(tag)
    # (the GC-Net key list was captured at max_disp = 64: the frozen disparity-sample tensor's length differs, not its name)
    strip = lambda ks: set(k for k in ks if not k.startswith("disp_predictor.disp_regression.weight"))
    assert strip(got) == strip(expect), (sorted(got - expect)[:5], sorted(expect - got)[:5])
    assert any(k.startswith("disp_predictor.disp_regression.weight") for k in got)
    assert "losses" in cfg.model     # the training branch of the model needs them


def test_flat_gradients_views():
    """dist_utils.FlatGradients("accumulate") without a process group: every grad is a view into one buffer, autograd accumulates
    into the views, zero_() clears them in place and re-attaches views an optimizer dropped."""
    from densematchingbenchmark_amd.dist_utils import FlatGradients
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    flat = FlatGradients(model, mode="accumulate")
    assert flat.attached() and flat.flat.numel() >= sum(p.numel() for p in model.parameters())
    model(torch.randn(4, 5)).sum().backward()
    assert flat.attached() and flat.flat.abs().sum().item() > 0
    ref = [p.grad.clone() for p in model.parameters()]
    model(torch.randn(4, 5)).sum().backward()                 # accumulates in place
    assert all(not torch.equal(p.grad, r) for p, r in zip(model.parameters(), ref))
    model.zero_grad(set_to_none=True)
    assert not flat.attached()
    flat.zero_()
    assert flat.attached() and flat.flat.abs().sum().item() == 0


def test_flat_gradients_gather_mode():
    """The default mode (round 6): zero_() drops the gradients, backward leaves fresh tensors on the parameters (no accumulation
    launch per parameter), gather_() packs them into the buffer with one multi-tensor copy and turns every grad into its view;
    a parameter that took no part in the step contributes zeros."""
    from densematchingbenchmark_amd.dist_utils import FlatGradients
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    unused = torch.nn.Parameter(torch.ones(3))
    model.register_parameter("unused", unused)
    flat = FlatGradients(model)
    assert flat.mode == "gather" and not flat.attached()
    flat.flat.fill_(7.0)                                       # stale contents must not survive a gather
    x = torch.randn(4, 5)
    flat.zero_()
    assert all(p.grad is None for p in model.parameters())
    model(x).sum().backward()
    assert not flat.attached() and unused.grad is None
    ref = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in model.parameters()]
    flat.gather_()
    assert flat.attached()
    assert all(torch.equal(p.grad, r) for p, r in zip(model.parameters(), ref))
    covered = torch.zeros_like(flat.flat, dtype=torch.bool)
    for p, o in zip(flat.params, flat.offsets):
        covered[o:o + p.numel()] = True
    assert not covered.all() and float(flat.flat[~covered].abs().sum()) == 7.0 * int((~covered).sum())   # only alignment gaps keep old bytes
    model(x).sum().backward()                                  # a second pass without zero_() accumulates into the views
    assert flat.attached() and all(torch.allclose(p.grad, 2 * r) for p, r in zip(model.parameters(), ref))
    flat.zero_()
    assert all(p.grad is None for p in model.parameters())


def test_confidence_head_composed_with_upsampling_weights():
    """ops.conf_head_k8s4_weights (pure tensor algebra, CPU) against the definition: for interior pixels, the head's 3x3
    convolution of ConvTranspose3d(1, 1, 8, 4, 2)(c) equals the phase-wise 3x3 convolution of c with the composed weights
    (cmn/cmn.py:21-27 o aggregators/AcfNet.py:55-57,81-83)."""
    import torch.nn.functional as F
    from densematchingbenchmark_amd import ops
    g = torch.Generator().manual_seed(3)
    Dq, Hq, Wq, M = 3, 5, 6, 4
    c = torch.randn((1, 1, Dq, Hq, Wq), generator=g, dtype=torch.float64)
    w8 = torch.randn((1, 1, 8, 8, 8), generator=g, dtype=torch.float64)
    w1 = torch.randn((M, 4 * Dq, 3, 3), generator=g, dtype=torch.float64)
    hidden = F.conv2d(F.conv_transpose3d(c, w8, stride=4, padding=2).squeeze(1), w1, padding=1)      # [1, M, 4Hq, 4Wq]
    K = ops.conf_head_k8s4_weights(w1.float(), w8.float())
    assert K.shape == (16 * M, Dq, 3, 3) and K.dtype == torch.float32
    hq = F.conv2d(c.squeeze(1), K.double(), padding=1).reshape(1, 4, 4, M, Hq, Wq)                   # [by, bx, m]
    comp = hq.permute(0, 3, 4, 1, 5, 2).reshape(1, M, 4 * Hq, 4 * Wq)
    inner = (slice(None), slice(None), slice(1, -1), slice(1, -1))
    assert (comp[inner] - hidden[inner]).abs().max().item() <= 1e-5 * hidden.abs().max().item()
    # the outermost pixel ring is where the two differ (the head zero-pads the up-sampled volume): conf_ring_kernel's job
    assert (comp - hidden).abs().max().item() > 1e-3


def test_pfm_loader_matches_the_reference_loader(tmp_path):
    """disp_io.load_pfm / load_scene_flow_disp (drop-in for dmb/data/datasets/utils/load_disp.py:5-68) on three files whose
    bytes and expected contents -- as the REFERENCE's loader returns them -- are fixtures (oracle/gen_golden_pfm.py): gray
    little-endian, gray big-endian with a scale, colour; same array (rows flipped to top-down), same byte-order dtype, same
    scale; the reference's two error cases; and a write -> read round trip."""
    import numpy as np
    from densematchingbenchmark_amd import disp_io
    g = golden("pfm_files.npz")
    for name in ("gray_le", "gray_be", "color_le"):
        p = str(tmp_path / (name + ".pfm"))
        with open(p, "wb") as fp:
            fp.write(g[name + "_bytes"].tobytes())
        data, scale = disp_io.load_pfm(p)
        assert np.array_equal(data, g[name + "_data"]) and scale == float(g[name + "_scale"])
        assert data.dtype.str == str(g[name + "_dtype"])
        if name.startswith("gray"):
            assert np.array_equal(disp_io.load_scene_flow_disp(p), g[name + "_data"])
    bad = str(tmp_path / "bad.pfm")
    with open(bad, "wb") as fp:
        fp.write(b"P6\n2 2\n-1.0\n" + b"\0" * 16)
    with pytest.raises(Exception, match="Not a PFM file"):
        disp_io.load_pfm(bad)
    with open(bad, "wb") as fp:
        fp.write(b"Pf\n2x2\n-1.0\n" + b"\0" * 16)
    with pytest.raises(Exception, match="Malformed PFM header"):
        disp_io.load_pfm(bad)
    with pytest.raises(AssertionError):
        disp_io.load_scene_flow_disp(str(tmp_path / "x.png"))
    arr = np.arange(12, dtype=np.float32).reshape(3, 4) * 0.5
    rt = str(tmp_path / "rt.pfm")
    disp_io.write_pfm(rt, arr, 3.0, little_endian=False)
    back, s = disp_io.load_pfm(rt)
    assert np.array_equal(back, arr) and s == 3.0


REFERENCE_CONFIGS = os.path.join(os.environ.get("DMB_REFERENCE", "/root/reference"), "configs")
IN_SCOPE = ("PSMNet", "AcfNet", "StereoNet", "GCNet")


@pytest.mark.skipif(not os.path.isdir(REFERENCE_CONFIGS), reason="the reference tree is only present in the build container")
def test_every_in_scope_reference_config_loads_unchanged():
    """Drop-in boundary: the reference's OWN config files (read in place, not copied) go through Config.fromfile + build_model
    unchanged -- every file of the in-scope families, incl. the KITTI ones (configs/PSMNet/kitti_2015.py: 384x1248); the
    out-of-scope families refuse loudly instead of building something else."""
    from densematchingbenchmark_amd.modeling import build_model
    seen = 0
    for fam in sorted(os.listdir(REFERENCE_CONFIGS)):
        d = os.path.join(REFERENCE_CONFIGS, fam)
        if not os.path.isdir(d):
            continue
        for f in sorted(os.listdir(d)):
            if not f.endswith(".py"):
                continue
            cfg = Config.fromfile(os.path.join(d, f))
            if fam in IN_SCOPE:
                model = build_model(cfg, backbone=None)
                assert sum(p.numel() for p in model.parameters()) > 0
                assert list(cfg.data.eval.input_shape) in ([544, 960], [384, 1248]), (fam, f)
                seen += 1
            else:
                with pytest.raises(NotImplementedError):
                    build_model(cfg, backbone=None)
    assert seen >= 10


@pytest.mark.skipif(not os.path.isdir(REFERENCE_CONFIGS), reason="the reference tree is only present in the build container")
def test_reference_state_dicts_load_strictly_into_default_build_model():
    """``build_model(cfg)`` is the reference's ``build_model(cfg)`` (dmb/modeling/__init__.py:10): for every in-scope file of the
    reference's own configs/ tree the REFERENCE model's state_dict loads ``strict=True`` into this package's default build --
    backbone included (dmb/apis/inference.py:61-85).  Runs oracle/check_strict_load.py in a child process (it imports the
    reference with stubbed third-party modules, which must not leak into this interpreter)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, os.path.join(root, "oracle", "check_strict_load.py")], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert len(res) >= 10
    bad = {k: v for k, v in res.items() if v[1] is not None}
    assert not bad, bad
    assert res["configs/PSMNet/scene_flow.py"][0] == 517 and res["configs/AcfNet/scene_flow_adaptive.py"][0] == 548


def test_build_model_backbone_argument():
    """Default = the backbone the config names; ``None`` = the cost path alone; a config without the entry builds none."""
    from densematchingbenchmark_amd.modeling import build_model
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", "PSMNet", "scene_flow.py"))
    assert build_model(cfg).backbone is not None and build_model(cfg, backbone=None).backbone is None
    assert any(k.startswith("backbone.") for k in build_model(cfg).state_dict())
    assert build_model(_cfg()).backbone is None                  # no cfg.model.backbone entry: nothing to build
    with pytest.raises(AttributeError):
        build_model(_cfg(), backbone="hip")                      # "hip" insists on the entry
    gwc = Config.fromfile(os.path.join(root, "configs", "GwcNet", "scene_flow.py"))
    assert "backbone" not in gwc.model and build_model(gwc).backbone is None
    with pytest.raises(ValueError):
        build_model(cfg, backbone="torch")


def test_dmb_ops_namespace():
    """``from dmb.ops import GateRecurrent2dnoind`` (dmb/ops/__init__.py:1) keeps working with the package name swapped."""
    from densematchingbenchmark_amd.ops import GateRecurrent2dnoind
    from densematchingbenchmark_amd.spn import GateRecurrent2dnoind as G2
    assert GateRecurrent2dnoind is G2 and GateRecurrent2dnoind(True, False).horizontal is True


def test_binding_constants_match_the_header():
    import re
    text = open(_lib.HEADER_PATH).read()
    assert int(re.search(r"#define DMB_DECONV3D_WORKSPACE_BYTES (\d+)", text).group(1)) == _lib.DECONV3D_WORKSPACE_BYTES
    assert int(re.search(r"#define DMB_CONV_SINGLE_CHAIN (0x[0-9a-f]+)", text).group(1), 16) == _lib.CONV_SINGLE_CHAIN
    assert "ABI version" in text and "(8:" in text and _lib.ABI_VERSION == 8


def test_data_side_and_serving_api_refuse_host_tensors():
    """The data-side transforms and the graph policy have no CPU path either: host tensors raise, and the auto policy never asks
    for a graph on them."""
    from densematchingbenchmark_amd.data import Compose, Normalize, StereoPad
    from densematchingbenchmark_amd.graph_runner import wants_graph
    sample = dict(leftImage=torch.zeros(3, 8, 12), rightImage=torch.zeros(3, 8, 12))
    with pytest.raises(_lib.DmbLibraryError):
        Compose([StereoPad((8, 16)), Normalize(ops.IMAGENET_MEAN, ops.IMAGENET_STD)])(dict(sample))
    with pytest.raises(_lib.DmbLibraryError):
        ops.stereo_pad_normalize(torch.zeros(1, 3, 8, 12), (8, 16))
    assert not wants_graph(sample) and not wants_graph({})
    from densematchingbenchmark_amd.apis import init_model, is_image_file, is_pfm_file
    assert is_image_file("a/b.png") and is_pfm_file("x.pfm") and not is_image_file("x.pfm")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    m = init_model(os.path.join(root, "configs", "PSMNet", "scene_flow.py"), None, "cpu")
    assert m.backbone is not None and not m.training and m.cfg.model.max_disp == 192
    sd = {"module." + k: v for k, v in m.state_dict().items()}          # a DataParallel checkpoint as mmcv writes it
    m2 = init_model(os.path.join(root, "configs", "PSMNet", "scene_flow.py"), {"state_dict": sd}, "cpu")
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))


def test_branch_overlap_auto_policy():
    """Default "auto": on for ONE pair of up to 544x960 / max_disp 192 at quarter resolution, off for batches; explicit values win."""
    class V:
        def __init__(self, *shape):
            self.shape = shape
    try:
        ops.set_branch_overlap("auto")
        assert ops.branch_overlap(V(1, 64, 48, 136, 240)) and ops.branch_overlap(V(1, 64, 16, 64, 128))
        assert not ops.branch_overlap(V(4, 64, 48, 136, 240)) and not ops.branch_overlap(V(2, 64, 48, 96, 312)) and not ops.branch_overlap()
        ops.set_branch_overlap(True)
        assert ops.branch_overlap(V(4, 64, 48, 136, 240)) is True
        ops.set_branch_overlap(False)
        assert ops.branch_overlap(V(1, 64, 16, 64, 128)) is False
    finally:
        ops.set_branch_overlap("auto")


def test_library_refuses_a_binary_built_from_other_sources(monkeypatch):
    """The library carries the sha256 of the sources it was linked from (dmb_build_id); the binding recomputes it from the sources
    next to the library and refuses a mismatch -- here simulated by a digest function that sees 'other' sources."""
    from densematchingbenchmark_amd import build
    lib = _lib.load()
    assert lib.dmb_build_id().decode() == build.sources_digest(dev=_lib.DEV_BUILD) and len(lib.dmb_build_id()) == 64
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(build, "sources_digest", lambda dev=False, defs="": "0" * 64)
    with pytest.raises(_lib.DmbLibraryError, match="built from other sources"):
        _lib.load()
    monkeypatch.undo()
    assert _lib.load() is not None


def test_graph_runner_flattens_nested_batches():
    from densematchingbenchmark_amd.graph_runner import _flatten, _rebuild
    a, b, c = torch.zeros(1), torch.ones(2), torch.full((3,), 2.0)
    batch = dict(leftFeature=(a, b), rightFeature=c, original_size=(540, 960), name="x")
    flat = _flatten(batch)
    assert [p for p, _ in flat] == ["/leftFeature/0", "/leftFeature/1", "/rightFeature"]
    again = _rebuild(batch, {p: t + 1 for p, t in flat})
    assert isinstance(again["leftFeature"], tuple) and torch.equal(again["leftFeature"][1], b + 1) and again["original_size"] == (540, 960)


def test_baseline_cfg0_config_builds():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", "PSMNet", "baseline_cfg0_256x512_d64.py"))
    assert cfg.model.max_disp == 64 and cfg.model.cost_processor.cost_computation.max_disp == 16 and list(cfg.data.eval.input_shape) == [256, 512]
    from densematchingbenchmark_amd.modeling import build_model
    m = build_model(cfg)
    assert m.backbone is not None and m.disp_predictor.max_disp == 64


def test_imread_keeps_16_bit_disparity_pngs(tmp_path):
    """KITTI ground truth is a 16-bit PNG (disparity * 256, apis/inference.py:36-46 divides by disp_div_factor): the decoder
    wrapper must hand back the 16-bit values, as imageio does, not an 8-bit conversion."""
    from PIL import Image
    from densematchingbenchmark_amd.apis import load_disp
    from densematchingbenchmark_amd.data import imread
    a = (np.arange(12, dtype=np.uint16).reshape(3, 4) * 5000)
    Image.fromarray(a).save(str(tmp_path / "k.png"))
    got = imread(str(tmp_path / "k.png"))
    assert got.dtype == np.uint16 and got.shape == (3, 4) and np.array_equal(got, a)
    d = load_disp({"left_disp_map_path": str(tmp_path / "k.png")}, "left_disp_map_path", 256.0)
    assert d.dtype == np.float32 and d.shape == (3, 4) and np.allclose(d, a.astype(np.float32) / 256.0)
    rgb = (np.arange(36, dtype=np.uint8).reshape(3, 4, 3))
    Image.fromarray(rgb).save(str(tmp_path / "c.png"))
    assert np.array_equal(imread(str(tmp_path / "c.png")), rgb)


def test_gradient_carry_scope_bookkeeping():
    """train_fn's carry scope without any kernel: the registry exists only inside a scope, nested scopes share it, the latest alias
    of a tensor is found through the chain, only gradient-requiring contiguous FP32 tensors are carried, a second thread has its
    own (no) scope, and leaving the scope drops every reference."""
    import threading
    from densematchingbenchmark_amd.modeling.stereo.layers import train_fn
    x = torch.randn(2, 3, requires_grad=True)
    plain = torch.randn(2, 3)
    assert train_fn._carry_plan(x, None) == (None, x, None, False, False)          # no scope: nothing is carried
    with train_fn.carry_scope():
        reg = train_fn._registry()
        assert reg == {}
        with train_fn.carry_scope():
            assert train_fn._registry() is reg                                       # nested: the outermost registry
        assert train_fn._registry() is reg
        r, x1, s1, cx, cs = train_fn._carry_plan(x, plain)
        assert r is reg and x1 is x and s1 is plain and cx and not cs               # a skip without gradient is not carried
        a1 = x.view_as(x)
        reg[id(x)] = (x, a1)
        a2 = a1.view_as(a1)
        reg[id(a1)] = (a1, a2)
        assert train_fn._latest(reg, x) is a2 and train_fn._latest(reg, a1) is a2    # consumers chain onto the latest alias
        r, x2, s2, cx, cs = train_fn._carry_plan(plain, x)
        assert x2 is plain and s2 is a2 and not cx and cs
        r, x3, s3, cx, cs = train_fn._carry_plan(x, x)
        assert x3 is a2 and s3 is a2 and cx and not cs                               # the same tensor twice: carried once
        assert train_fn._carry_plan(x.t(), None)[3] is False                         # not contiguous: not carried
        with torch.no_grad():
            assert train_fn._carry_plan(x, None)[0] is None
        train_fn.set_gradient_carry(False)
        try:
            assert train_fn._carry_plan(x, None)[0] is None
        finally:
            train_fn.set_gradient_carry(True)
        seen = []
        t = threading.Thread(target=lambda: seen.append(train_fn._registry()))
        t.start()
        t.join()
        assert seen == [None]
    assert train_fn._registry() is None and reg == {}
