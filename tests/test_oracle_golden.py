"""Pin the CPU oracle (oracle/dmb_oracle.py) against vectors produced by the REAL reference
(tests/golden/*.npz, written by oracle/gen_golden.py in the build container).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import dmb_oracle as O
from tests._util import golden, maxdiff, rand, sha


def test_disp_index_truncation():
    # max_disp=6, dilation=2 -> [0, 2, 5] (int() of an FP32 linspace)
    assert O.disp_index_list(6, 0, 2) == [0, 2, 5]
    assert O.disp_index_list(5, -2, 2) == [-2, 0, 2]
    assert O.disp_index_list(48, 0, 1) == list(range(48))


def test_volumes_known_answer_and_seeded():
    g = golden("volumes.npz")
    L = torch.arange(1, 13, dtype=torch.float32).view(1, 1, 3, 4)
    R = torch.arange(13, 25, dtype=torch.float32).view(1, 1, 3, 4)
    assert np.array_equal(O.cat_fms(L, R, 5, -2, 2).numpy(), g["ka_cat"])
    assert np.array_equal(O.dif_fms(L, R, 5, -2, 2).numpy(), g["ka_dif"])
    # the values the reference's own test prints (tests/.../test_cat_fms.py:30-40)
    assert g["ka_cat"][0, 0, :, 0].tolist() == [[1, 2, 0, 0], [1, 2, 3, 4], [0, 0, 3, 4]]
    assert g["ka_cat"][0, 1, :, 0].tolist() == [[15, 16, 0, 0], [13, 14, 15, 16], [0, 0, 13, 14]]
    assert g["ka_dif"][0, 0, :, 0].tolist() == [[-14, -14, 0, 0], [-12, -12, -12, -12], [0, 0, -10, -10]]
    for i, row in enumerate(g["cases"]):
        shape, (md, sd, dil, seed) = tuple(int(v) for v in row[:4]), (int(v) for v in row[4:])
        a, b = rand(shape, seed), rand(shape, seed + 1000)
        c, d = O.cat_fms(a, b, md, sd, dil), O.dif_fms(a, b, md, sd, dil)
        assert sha(c) == str(g["cat_sha_%d" % i]) and sha(d) == str(g["dif_sha_%d" % i])  # bit-exact
        if "cat_%d" % i in g:
            assert np.array_equal(c.numpy(), g["cat_%d" % i]) and np.array_equal(d.numpy(), g["dif_%d" % i])


def _fast_case(row):
    shape, D, seed = tuple(int(v) for v in row[:4]), int(row[4]), int(row[5])
    a, b = rand(shape, seed), rand(shape, seed + 1000)
    g = torch.Generator().manual_seed(seed + 2000)
    ds = torch.rand((shape[0], D, shape[2], shape[3]), generator=g) * shape[3] * 0.6 - 2.0
    return a, b, ds


def test_fast_mode_volumes_vs_reference():
    """The sample-based builders (cat_fms.py:51-82, dif_fms.py:49-86): the oracle's own statement of the sampler against
    the reference's outputs on this torch, bit for bit -- its test's case (test_cat_fms.py:52-80) and seeded per-pixel samples."""
    g = golden("fast_volumes.npz")
    L = torch.arange(1, 13, dtype=torch.float32).view(1, 1, 3, 4)
    R = torch.arange(13, 25, dtype=torch.float32).view(1, 1, 3, 4)
    assert np.array_equal(O.fast_cat_fms(L, R, 5, -2, 2).numpy(), g["ka_cat"])
    ka_samples = torch.linspace(-2, 2, 3).repeat(1, 3, 4, 1).permute(0, 3, 1, 2).contiguous()
    assert np.array_equal(O.fast_cat_fms(L, R, 5, -2, 2, ka_samples).numpy(), g["ka_cat_samples"])
    assert np.array_equal(g["ka_cat"], g["ka_cat_samples"])
    assert np.array_equal(O.fast_dif_fms(L, R, 5, -2, 2).numpy(), g["ka_dif"])
    assert not np.array_equal(g["ka_cat"], golden("volumes.npz")["ka_cat"])      # not the default builder's volume
    for i, row in enumerate(g["cases"]):
        a, b, ds = _fast_case(row)
        c, d = O.fast_cat_fms(a, b, disp_sample=ds), O.fast_dif_fms(a, b, disp_sample=ds)
        assert sha(c) == str(g["cat_sha_%d" % i]) and sha(d) == str(g["dif_sha_%d" % i])
        if "cat_%d" % i in g:
            assert np.array_equal(c.numpy(), g["cat_%d" % i]) and np.array_equal(d.numpy(), g["dif_%d" % i])
        assert maxdiff(O.fast_dif_fms(a, b, disp_sample=ds, normalize=True, p=1.0), g["dif_norm1_%d" % i]) <= 1e-5
        assert maxdiff(O.fast_dif_fms(a, b, disp_sample=ds, normalize=True, p=2.0), g["dif_norm2_%d" % i]) <= 1e-5
        c = O.fast_cat_fms(a, b, 24, -3, 2)
        assert sha(c) == str(g["cat_default_sha_%d" % i])
        assert np.array_equal(c[:, :, ::5, 1::3].numpy(), g["cat_default_rows_%d" % i])


def test_predictors():
    g = golden("predictors.npz")
    ones = torch.ones(1, 5, 2, 2)
    kw = dict(max_disp=9, start_disp=-4, dilation=2, alpha=1.0)
    assert maxdiff(O.soft_argmin(ones, **kw), g["ka_soft"]) <= 1e-7
    assert maxdiff(O.faster_soft_argmin(ones, **kw), g["ka_faster"]) <= 1e-7
    assert np.array_equal(O.local_soft_argmin(ones, radius=2, **kw)[0].numpy(), g["ka_local"])
    assert float(g["ka_local"].max()) == -2.0 and abs(float(g["ka_soft"].max())) < 1e-6
    for tag in ("flat", "peaked", "d192"):
        D, seed = (int(v) for v in g[tag + "_meta"])
        cost = rand((2, D, 6, 10), seed, float(g[tag + "_gain"][0]))
        assert maxdiff(O.soft_argmin(cost, D), g[tag + "_soft"]) <= 2e-5
        assert maxdiff(O.faster_soft_argmin(cost, D), g[tag + "_faster"]) <= 2e-5
        disp, idx = O.local_soft_argmin(cost, D, 2)
        assert np.array_equal(idx.numpy(), g[tag + "_argmax"])          # index path bit-exact
        assert maxdiff(disp, g[tag + "_local"]) <= 2e-5
        # the FP64 truth is where the reference's two FP32 variants meet (within ~1e-4 at D=192)
        assert maxdiff(O.soft_argmin_f64(cost, D).float(), g[tag + "_faster"]) <= 1.5e-4
    cost = rand((1, 12, 4, 6), 204, 5.0)
    assert maxdiff(O.soft_argmin(cost, 24, -6, 2, 0.7), g["dil_soft"]) <= 1e-5
    assert maxdiff(O.local_soft_argmin(cost, 24, 3, -6, 2, 2, 0.7)[0], g["dil_local"]) <= 1e-5
    assert maxdiff(O.soft_argmin(cost, 12, 0, 1, 0.5, normalize=False), g["nonorm_soft"]) <= 1e-4
    samp = rand((1, 12, 4, 6), 205, 10.0)
    assert maxdiff(O.soft_argmin(cost, 12, disp_sample=samp), g["sampled_soft"]) <= 1e-5


def test_aggregators():
    g = golden("aggregators.npz")
    raw = rand((1, 64, 8, 16, 32), 301)
    p = O.random_params_psm(seed=0, classif_gain=10.0)
    costs = O.psm_aggregator(raw, p, 32)
    for c, k in zip(costs, ("psm_cost3", "psm_cost2", "psm_cost1")):
        assert maxdiff(c[:, ::4, ::8, :], g[k]) <= 1e-5
    for c, k in zip(costs, ("psm_disp3", "psm_disp2", "psm_disp1")):
        assert maxdiff(O.faster_soft_argmin(c, 32), g[k]) <= 1e-5
    assert np.ptp(g["psm_cost3"]) > 1.0  # the fixture is peaked, not the degenerate default-init volume (ptp ~ 8e-3)
    # hourglass wiring incl. presqu/postsqu cross links
    c0 = O.conv3d_unit(O.conv3d_unit(raw, p, "dres0.0", relu=True), p, "dres0.1", relu=True)
    c0 = O.conv3d_unit(O.conv3d_unit(c0, p, "dres1.0", relu=True), p, "dres1.1") + c0
    o1, pre1, post1 = O.hourglass(c0, None, None, p, "dres2")
    assert maxdiff(o1[:, ::8], g["hg_out"]) <= 1e-5 and maxdiff(pre1[:, ::16], g["hg_pre"]) <= 1e-5
    assert maxdiff(post1[:, ::16], g["hg_post"]) <= 1e-5
    o2, pre2, post2 = O.hourglass(o1 + c0, pre1, post1, p, "dres3")
    assert maxdiff(o2[:, ::8], g["hg2_out"]) <= 1e-5 and maxdiff(pre2[:, ::16], g["hg2_pre"]) <= 1e-5
    assert maxdiff(post2[:, ::16], g["hg2_post"]) <= 1e-5

    p = O.random_params_psm(seed=1, classif_gain=10.0, acf=True)
    costs = O.acf_aggregator(raw, p, 32)
    for c, k in zip(costs, ("acf_cost3", "acf_cost2", "acf_cost1")):
        assert maxdiff(c[:, ::4, ::8, :], g[k]) <= 2e-5
    for i, c in enumerate(costs):
        hp = {"h." + k[len("confp_%d_" % i):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("confp_%d_" % i)}
        conf, conf_cost = O.conf_head(c, hp, "h")
        assert maxdiff(conf_cost, g["conf_cost_%d" % i]) <= 2e-5 and maxdiff(conf, g["conf_%d" % i]) <= 1e-5

    sp = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("snp_")}
    cost = O.stereonet_aggregator(rand((2, 32, 6, 10, 20), 302), sp)[0]
    assert maxdiff(cost, g["sn_cost"]) <= 1e-5


def test_psmnet_path_cfg1():
    """BASELINE config #1 (PSMNet, 256x512, max_disp 64) through the reference's own builders."""
    g = golden("psmnet_path_cfg1.npz")
    p = O.with_prefix(O.random_params_psm(seed=2, classif_gain=10.0), "cost_processor.aggregator.")
    lf, rf = rand((1, 32, 64, 128), 401), rand((1, 32, 64, 128), 402)
    disps, costs = O.psmnet_path(lf, rf, p, 64)
    # same arithmetic on the same library as the fixture's generator; the bound leaves room for torch's thread-count
    # dependent summation order (2.3e-5 between 1 and 8 threads; 2.1e-5 seen on a 256-thread host)
    for i, (d, c) in enumerate(zip(disps, costs)):
        assert maxdiff(d[:, :, ::2, ::2], g["disp%d" % (3 - i)]) <= 4e-5
        assert maxdiff(c[:, ::8, ::32, :], g["cost%d_rows" % (3 - i)]) <= 2e-5


def test_evaluation():
    g = golden("evaluation.npz")
    gen = torch.Generator().manual_seed(501)
    gt = torch.rand((3, 1, 20, 32), generator=gen) * 220 - 10
    est = gt + torch.randn((3, 1, 20, 32), generator=gen) * 3
    gt[2] = -1.0
    crop = O.remove_padding(est, (17, 30))
    assert list(crop.shape) == g["cropped_shape"].tolist() and sha(crop) == str(g["cropped_sha"])
    for b in range(3):
        e = O.calc_error(O.remove_padding(est[b:b + 1], (17, 30)), O.remove_padding(gt[b:b + 1], (17, 30)), 0, 192)
        got = np.array([e[k] for k in ("epe", "1px", "2px", "3px", "5px")])
        assert np.allclose(got, g["img%d" % b], rtol=1e-6, atol=1e-6)
    assert g["img2"].tolist() == [0, 0, 0, 0, 0]


def _model_params(cfg_rel, seed, tweak=None):
    """Seeded parameters under the model-level names, from the package's generator applied to OUR module tree (the
    same call gen_golden.py applies to the reference's module tree: identical state-dict names => identical draws)."""
    import os
    from densematchingbenchmark_amd import synthetic
    from densematchingbenchmark_amd.config import Config
    from densematchingbenchmark_amd.modeling import build_model
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", cfg_rel))
    if tweak:
        tweak(cfg)
    model = build_model(cfg, backbone=None).eval()
    synthetic.init_params_(model, seed=seed, classif_gain=10.0)
    return cfg, model


def _acf32(cfg):
    cfg.model.max_disp = 32
    cfg.model.cost_processor.cost_computation.max_disp = 8
    cfg.model.cost_processor.cost_aggregator.max_disp = 32
    cfg.model.disp_predictor.max_disp = 32
    cfg.model.cmn.in_planes = 32


def test_acfnet_path_vs_reference():
    g = golden("acfnet_path.npz")
    cfg, model = _model_params("AcfNet/scene_flow_adaptive.py", 5, _acf32)
    p = {k: v.clone() for k, v in model.state_dict().items()}
    lf, rf = rand((2, 32, 16, 32), 411), rand((2, 32, 16, 32), 412)
    disps, costs, confs = O.acfnet_path(lf, rf, p, 32)
    for i in range(3):
        assert maxdiff(disps[i], g["disp%d" % (3 - i)]) <= 2e-5
        assert maxdiff(confs[i], g["conf%d" % (3 - i)]) <= 1e-5
        assert maxdiff(costs[i][:, ::4, ::8, :], g["cost%d_rows" % (3 - i)]) <= 2e-5
        assert maxdiff(1.0 * (1 - confs[i]) + 1.0, g["var%d" % (3 - i)]) <= 1e-5   # cmn.py:67: alpha*(1-conf)+beta


def test_stereonet_path_vs_reference():
    g = golden("stereonet_path.npz")
    cfg, model = _model_params("StereoNet/scene_flow_8x_2stage.py", 6)
    p = {k: v.clone() for k, v in model.state_dict().items()}
    lf, rf = rand((2, 32, 20, 36), 421), rand((2, 32, 20, 36), 422)
    disps, costs = O.stereonet_path(lf, rf, p, 192)
    assert maxdiff(costs[0], g["cost"]) <= 1e-5 and maxdiff(disps[0], g["disp"]) <= 2e-5
    assert g["disp"].shape == (2, 1, 20, 36) and g["cost"].shape == (2, 24, 20, 36)


def test_psmnet_backbone_and_end_to_end_vs_reference():
    """"Next" row: the oracle's backbone restatement against the reference's PSMNetBackbone, and the
    whole reference model (BASELINE configs[0]: 256x512, max_disp 64) against backbone -> path in the oracle."""
    import os
    from densematchingbenchmark_amd import synthetic
    from densematchingbenchmark_amd.config import Config
    from densematchingbenchmark_amd.modeling import build_model
    from densematchingbenchmark_amd.modeling.stereo.backbones import PSMNetBackbone
    g = golden("psmnet_backbone.npz")
    bb = PSMNetBackbone(3, True).eval()
    synthetic.init_params_(bb, seed=8, classif_gain=1.0)
    p = {"backbone." + k: v.clone() for k, v in bb.state_dict().items()}
    f = O.psmnet_backbone(rand((1, 3, 256, 512), 441), p)
    assert maxdiff(f[:, :, ::2, ::2], g["feat"]) <= 2e-6
    assert abs(float(f.std()) - g["feat_stats"][1]) <= 1e-5 and g["feat_stats"][2] > 0.1   # not a dead network

    g = golden("psmnet_e2e_cfg1.npz")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", "PSMNet", "scene_flow.py"))
    cfg.model.max_disp = 64
    cfg.model.cost_processor.cost_computation.max_disp = 16
    cfg.model.cost_processor.cost_aggregator.max_disp = 64
    cfg.model.disp_predictor.max_disp = 64
    cfg.model.backbone = dict(type="PSMNet", in_planes=3)
    model = build_model(cfg, backbone="hip").eval()
    assert sum(p_.numel() for p_ in model.parameters()) == int(g["n_params"][0]) == 5225024
    synthetic.init_params_(model, seed=9, classif_gain=10.0)
    p = {k: v.clone() for k, v in model.state_dict().items()}
    li, ri = rand((1, 3, 256, 512), 451), rand((1, 3, 256, 512), 452)
    disps, _ = O.psmnet_path(O.psmnet_backbone(li, p), O.psmnet_backbone(ri, p), p, 64)
    for i, d in enumerate(disps):   # (bit-equal on the fixture generator's host; 2.1e-5 on a host whose mkldnn picks other kernels)
        assert maxdiff(d[:, :, ::2, ::2], g["disp%d" % (3 - i)]) <= 4e-5
    want = set(str(s) for s in golden("state_dict_keys.npz")["psmnet_backbone"])
    got = set("%s %s" % (k, tuple(v.shape)) for k, v in model.state_dict().items() if k.startswith("backbone"))
    assert got == want and len(got) == 363


def test_stereonet_refinement_vs_reference():
    """"Next" row: oracle restatement of the edge-aware refinement cascade against the reference's
    StereoNetRefinement, and the drop-in module's parameter names against the reference's."""
    import os
    from densematchingbenchmark_amd import synthetic
    from densematchingbenchmark_amd.config import Config
    from densematchingbenchmark_amd.modeling import build_model
    from densematchingbenchmark_amd.modeling.stereo.disp_refinement import StereoNetRefinement
    g = golden("stereonet_refinement.npz")
    rf = StereoNetRefinement(in_planes=4, batch_norm=True, num=2).eval()
    synthetic.init_params_(rf, seed=10, classif_gain=1.0)
    p = {"disp_refinement." + k: v.clone() for k, v in rf.state_dict().items()}
    gen = torch.Generator().manual_seed(461)
    coarse = torch.rand((2, 1, 24, 40), generator=gen) * 4.0
    outs = O.stereonet_refinement([coarse], rand((2, 3, 192, 320), 462), p, num=2)
    assert len(outs) == 3
    for i, d in enumerate(outs):
        assert maxdiff(d[:, :, ::2, ::2], g["refined%d" % i]) <= 1e-5
    assert g["residual_stats"].min() > 0.5      # the blocks really change the map

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", "StereoNet", "scene_flow_8x_2stage.py"))
    cfg.model.disp_refinement = dict(type="StereoNet", in_planes=4, num=1)
    model = build_model(cfg, backbone=None)
    want = set(str(s) for s in golden("state_dict_keys.npz")["stereonet_refinement"])
    got = set("%s %s" % (k, tuple(v.shape)) for k, v in model.state_dict().items() if k.startswith("disp_refinement"))
    assert got == want and len(got) == 81


def test_stereonet_end_to_end_vs_reference():
    """Whole reference StereoNet (scene_flow_8x_2stage) against backbone -> path -> refinement in the oracle."""
    import os
    from densematchingbenchmark_amd import synthetic
    from densematchingbenchmark_amd.config import Config
    from densematchingbenchmark_amd.modeling import build_model
    g = golden("stereonet_e2e.npz")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", "StereoNet", "scene_flow_8x_2stage.py"))
    cfg.model.backbone = dict(type="StereoNet", in_planes=3, downsample_num=3, residual_num=6)
    cfg.model.disp_refinement = dict(type="StereoNet", in_planes=4, num=1)
    model = build_model(cfg, backbone="hip").eval()
    assert sum(p_.numel() for p_ in model.parameters()) == int(g["n_params"][0])
    synthetic.init_params_(model, seed=12, classif_gain=10.0)
    p = {k: v.clone() for k, v in model.state_dict().items()}
    li, ri = rand((1, 3, 192, 320), 481), rand((1, 3, 192, 320), 482)
    lf, rf = O.stereonet_backbone(li, p), O.stereonet_backbone(ri, p)
    assert maxdiff(lf, g["left_feature"]) <= 1e-5
    disps, _ = O.stereonet_path(lf, rf, p, 192)
    outs = O.stereonet_refinement(disps, li, p, num=1)
    for i, d in enumerate(outs):
        assert maxdiff(d, g["disp%d" % i]) <= 5e-5
    want = set(str(s) for s in golden("state_dict_keys.npz")["stereonet_backbone"])
    got = set("%s %s" % (k, tuple(v.shape)) for k, v in model.state_dict().items() if k.startswith("backbone"))
    assert got == want and len(got) == 80


def test_gcnet_vs_reference():
    """the oracle's GC-Net restatement (aggregator; backbone -> cat volume at 1/2 -> aggregator -> soft-argmin)
    against the reference's GCAggregator and whole model, and the drop-in's parameter names against the reference's."""
    import os
    from densematchingbenchmark_amd import synthetic
    from densematchingbenchmark_amd.config import Config
    from densematchingbenchmark_amd.modeling import build_model
    from densematchingbenchmark_amd.modeling.stereo.cost_processors.aggregators import GCAggregator
    g = golden("gcnet_aggregator.npz")
    ga = GCAggregator(max_disp=32, in_planes=64, batch_norm=True).eval()
    synthetic.init_params_(ga, seed=13, classif_gain=30.0)
    p = {k: v.clone() for k, v in ga.state_dict().items()}
    cost = O.gc_aggregator(rand((1, 64, 16, 16, 32), 491), p)[0]
    assert cost.shape == (1, 32, 32, 64)
    assert maxdiff(cost[:, ::2, ::2, ::2], g["cost"]) <= 2e-5 and g["cost_stats"][1] > 0.5

    g = golden("gcnet_e2e.npz")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", "GCNet", "scene_flow.py"))
    cfg.model.max_disp = 64
    cfg.model.cost_processor.cost_computation.max_disp = 32
    cfg.model.cost_processor.cost_aggregator.max_disp = 64
    cfg.model.disp_predictor.max_disp = 64
    model = build_model(cfg, backbone="hip").eval()
    assert sum(p_.numel() for p_ in model.parameters()) == int(g["n_params"][0])
    synthetic.init_params_(model, seed=14, classif_gain=30.0)
    p = {k: v.clone() for k, v in model.state_dict().items()}
    li, ri = rand((1, 3, 64, 128), 492), rand((1, 3, 64, 128), 493)
    lf, rf = O.gcnet_backbone(li, p), O.gcnet_backbone(ri, p)
    assert maxdiff(lf[:, ::2], g["left_feature"]) <= 1e-5
    disps, _ = O.gcnet_path(lf, rf, p, 64)
    assert maxdiff(disps[0], g["disp"]) <= 5e-5
    want = set(str(s) for s in golden("state_dict_keys.npz")["gcnet"])
    got = set("%s %s" % (k, tuple(v.shape)) for k, v in model.state_dict().items())
    assert got == want and len(got) == 216


def test_losses_vs_reference():
    """SURVEY 8-f3 (first part): the oracle's restatement of StereoFocalLoss + LaplaceDisp2Prob, ConfidenceNllLoss and
    DispSmoothL1Loss against the reference's values and autograd gradients."""
    g = golden("losses.npz")
    gt = torch.from_numpy(g["gt"])
    for tag, coef, var in (("a", 0.0, 1.2), ("b", 5.0, torch.from_numpy(g["focal_b_var"]).requires_grad_(True))):
        cost = (rand((2, 48, 12, 20), 512) * 3.0).requires_grad_(True)
        loss = 0.7 * O.stereo_focal_loss(cost, gt, var, 48, 0, 1, coef)
        loss.backward()
        assert abs(float(loss.detach()) - g["focal_%s_loss" % tag][0]) <= 1e-5 * abs(g["focal_%s_loss" % tag][0])
        assert maxdiff(cost.grad, g["focal_%s_gcost" % tag]) <= 1e-6 * np.abs(g["focal_%s_gcost" % tag]).max() + 1e-9
        if tag == "b":
            assert maxdiff(var.grad, g["focal_b_gvar"]) <= 1e-5 * np.abs(g["focal_b_gvar"]).max()
    conf = (rand((2, 1, 12, 20), 513) * 2.0).requires_grad_(True)
    l = O.conf_nll_loss(conf, gt, 48)
    l.backward()
    assert abs(float(l.detach()) - g["conf_loss"][0]) <= 1e-6 and maxdiff(conf.grad, g["conf_grad"]) <= 1e-8
    est = (gt + rand((2, 1, 12, 20), 514) * 2.0).detach().requires_grad_(True)
    l = O.disp_smooth_l1_loss(est, gt, 48)
    l.backward()
    assert abs(float(l.detach()) - g["l1_loss"][0]) <= 1e-6 and maxdiff(est.grad, g["l1_grad"]) <= 1e-8


def _fingerprint(t):
    f = t.detach().double().reshape(-1)
    step = max(1, f.numel() // 16)
    return np.concatenate([[f.sum().item(), (f * f).sum().item()], f[::step][:16].numpy()])


def _check_fingerprints(g, prefix, grads, strip=""):
    """Every gradient of the oracle's training step against the fingerprint (sum, sum of squares, 16 strided samples) of the
    REAL reference's autograd gradient.  Both are torch CPU FP32 evaluations of the same graph up to operator fusion: 1e-4 of
    each tensor's scale (sum of squares: 2e-4)."""
    n = 0
    keys = [k for k in g.files if k.startswith(prefix + "_g_")]
    # gradients that are exactly zero in exact arithmetic (a bias in front of a batch-statistics BatchNorm) are rounding noise
    # in both evaluations: an absolute floor of 1e-6 of the largest gradient entry of the step
    zero = 1e-6 * max(np.abs(g[k][2:]).max() for k in keys)
    for key in keys:
        name = key[len(prefix) + 3:]
        t = grads[strip + name if name not in ("ref_fms", "tgt_fms") else name]
        want, got = g[key], _fingerprint(t)
        scale = max(np.abs(want[2:]).max(), np.sqrt(want[1] / max(1, t.numel())))
        assert np.abs(got[2:] - want[2:]).max() <= 1e-4 * scale + zero, name
        assert abs(got[1] - want[1]) <= 2e-4 * want[1] + zero * zero * t.numel(), name
        n += 1
    return n


def test_training_step_matches_reference_autograd():
    """oracle.psmnet_train_step / acfnet_train_step against ONE TRAINING ITERATION OF THE REFERENCE'S OWN MODULES (train() mode,
    its loss classes, torch.autograd; oracle/gen_golden.py section 4h): losses, every gradient, updated BatchNorm buffers."""
    g = golden("training.npz")
    gt = torch.rand((2, 1, 32, 96), generator=torch.Generator().manual_seed(13)) * 40.0 - 4.0
    lf, rf = rand((2, 32, 8, 24), 11), rand((2, 32, 8, 24), 12)
    p = O.with_prefix(O.random_params_psm(seed=7, classif_gain=4.0), "cost_processor.aggregator.")
    losses, grads, running = O.psmnet_train_step(lf, rf, p, 32, gt)
    assert np.allclose([float(x) for x in losses], g["psm_losses"], rtol=1e-5)
    assert _check_fingerprints(g, "psm", grads, "cost_processor.aggregator.") == 80
    assert np.allclose(running["cost_processor.aggregator.dres0.0.1.running_mean"].numpy(), g["psm_rm"], rtol=1e-5, atol=1e-7)
    assert np.allclose(running["cost_processor.aggregator.dres0.0.1.running_var"].numpy(), g["psm_rv"], rtol=1e-5, atol=1e-7)

    p = O.with_prefix(O.random_params_psm(seed=4, classif_gain=4.0, acf=True), "cost_processor.aggregator.")
    gq = torch.Generator().manual_seed(77)
    md, Cm = 32, 32 // 3
    for i in range(3):
        pre = "cmn.conf_heads.%d.conf_net." % i
        p[pre + "0.0.weight"] = (torch.rand((Cm, md, 3, 3), generator=gq) * 2 - 1) / (md * 9) ** 0.5
        p[pre + "0.1.weight"] = 0.5 + torch.rand(Cm, generator=gq)
        p[pre + "0.1.bias"] = (torch.rand(Cm, generator=gq) - 0.5) * 0.2
        p[pre + "0.1.running_mean"], p[pre + "0.1.running_var"] = torch.zeros(Cm), torch.ones(Cm)
        p[pre + "1.weight"] = (torch.rand((1, Cm, 1, 1), generator=gq) * 2 - 1) / Cm ** 0.5
    lf, rf = rand((2, 32, 8, 24), 41), rand((2, 32, 8, 24), 42)
    gt = torch.rand((2, 1, 32, 96), generator=torch.Generator().manual_seed(43)) * 40.0 - 4.0
    losses, grads, _ = O.acfnet_train_step(lf, rf, p, 32, gt, adaptive=True)
    assert sorted(losses) == [str(k) for k in g["acf_loss_keys"]]
    assert np.allclose([float(losses[k]) for k in sorted(losses)], g["acf_losses"], rtol=1e-5)
    assert _check_fingerprints(g, "acf", grads) == 102


def test_backbone_training_matches_reference_autograd():
    """oracle.psmnet_backbone_train_step against the reference's own PSMNetBackbone in train() mode (two views, per-view
    BatchNorm statistics): features, every parameter gradient, the updated buffers of the first BatchNorm."""
    from densematchingbenchmark_amd import synthetic
    from densematchingbenchmark_amd.modeling.stereo.backbones import PSMNetBackbone
    g = golden("training.npz")
    bb = PSMNetBackbone(3, True)
    synthetic.init_params_(bb, seed=8, classif_gain=1.0)
    p = {"backbone." + k: v.clone() for k, v in bb.state_dict().items()}
    li, ri = rand((2, 3, 256, 320), 81), rand((2, 3, 256, 320), 82)
    dl, dr = rand((2, 32, 64, 80), 83), rand((2, 32, 64, 80), 84)
    (fl, fr), grads, running = O.psmnet_backbone_train_step(li, ri, p, dl, dr)
    want = g["bb_feat"]
    for f, w in zip((fl, fr), want):
        assert np.abs(_fingerprint(f)[2:] - w[2:]).max() <= 1e-5
    assert _check_fingerprints(g, "bb", grads) == len(grads)
    assert np.allclose(running["backbone.firstconv.0.1.running_mean"].numpy(), g["bb_rm"], rtol=1e-5, atol=1e-7)
    assert np.allclose(running["backbone.firstconv.0.1.running_var"].numpy(), g["bb_rv"], rtol=1e-5, atol=1e-7)


def test_stereonet_training_matches_reference_autograd():
    """oracle.stereonet_e2e_train_step against one training iteration of the reference's WHOLE StereoNet model (build_model(cfg, backbone=None)
    .train(), its own loss evaluator and autograd; gen_golden.py section 4h)."""
    from densematchingbenchmark_amd import synthetic
    from densematchingbenchmark_amd.config import Config
    from densematchingbenchmark_amd.modeling import build_model
    import os
    g = golden("training.npz")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", "StereoNet", "scene_flow_8x_2stage.py"))
    cfg.model.disp_refinement = dict(type="StereoNet", in_planes=4, num=1)
    cfg.model.backbone = dict(type="StereoNet", in_planes=3)
    model = build_model(cfg, backbone="hip")          # parameter container only: same names / shapes as the reference's model
    synthetic.init_params_(model, seed=12, classif_gain=4.0)
    p = {k: v.clone() for k, v in model.state_dict().items()}
    li, ri = rand((2, 3, 64, 96), 97), rand((2, 3, 64, 96), 98)
    gt = torch.rand((2, 1, 64, 96), generator=torch.Generator().manual_seed(99)) * 30.0 + 0.5
    losses, grads, _ = O.stereonet_e2e_train_step(li, ri, p, 192, gt)
    assert [str(k) for k in g["sn_loss_keys"]] == ["l1_loss_lvl0", "l1_loss_lvl1"]
    assert np.allclose([float(x) for x in losses], g["sn_losses"], rtol=1e-5)
    grads = {k: v for k, v in grads.items() if v is not None}
    n = _check_fingerprints(g, "sn", grads)
    assert n == len(grads) and n >= 100


# ------------------------------------------------------------------------------------------------- BASELINE sizes
def _fullsize_model(cfg_rel, seed):
    import os
    from densematchingbenchmark_amd import synthetic
    from densematchingbenchmark_amd.config import Config
    from densematchingbenchmark_amd.modeling import build_model
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    model = build_model(Config.fromfile(os.path.join(root, "configs", cfg_rel)), backbone=None).eval()   # parameter container only
    synthetic.init_params_(model, seed=seed, classif_gain=10.0)
    return {k: v.clone() for k, v in model.state_dict().items()}


def test_oracle_at_baseline_size_matches_reference_psmnet_and_stereonet():
    """The oracle's FP32 path at the BASELINE sizes against the reference's own outputs there
    (tests/golden/fullsize_*.npz, oracle/gen_golden_fullsize.py): PSMNet 544x960 / max_disp 192 (pair 0 of the bench batch)
    and StereoNet-8x 384x1248.  Same arithmetic on the same library (torch CPU): bit-level agreement is expected, the
    asserts allow a thread-count dependent summation order (measured 2e-5 between 1 and 8 threads)."""
    from densematchingbenchmark_amd import synthetic
    sub = (slice(None), slice(None), slice(3, None, 8), slice(5, None, 8))
    crows = (slice(None), slice(7, None, 48), slice(11, None, 136), slice(None))
    with torch.no_grad():
        g = golden("fullsize_stereonet.npz")
        p = _fullsize_model("StereoNet/scene_flow_8x_2stage.py", 6)
        lf, rf = synthetic.feature_pair(0, 32, 48, 156)
        disps, costs = O.stereonet_path(lf, rf, p, 192)
        assert maxdiff(disps[0], g["disp"]) <= 2e-5 and maxdiff(costs[0][:, :, 1::2, :], g["cost"]) <= 1e-5
        g = golden("fullsize_psmnet.npz")
        p = _fullsize_model("PSMNet/scene_flow.py", 0)
        lf, rf = synthetic.feature_pair(0, 32, 136, 240)
        disps, costs = O.psmnet_path(lf, rf, p, 192)
        for lvl in range(3):
            assert maxdiff(disps[lvl][sub], g["pair0_disp%d" % (3 - lvl)]) <= 5e-5
            assert maxdiff(costs[lvl][crows], g["pair0_cost%d_rows" % (3 - lvl)]) <= 2e-5


def test_oracle_regression_tail_at_the_ends_of_the_range():
    """Round-3 fixture (oracle/gen_golden_fullsize.py round3): trilinear x4 up-sampling (PSMNet.py:74-93) + the reference's two
    soft-argmin modules on a volume with ground-truth-like peaks near disparity 5 and 185 and costs spanning +-12."""
    from densematchingbenchmark_amd import synthetic
    g = golden("fullsize_regression_ends.npz")
    sub = (slice(None), slice(None), slice(3, None, 8), slice(5, None, 8))
    crows = (slice(None), slice(7, None, 48), slice(11, None, 136), slice(None))
    q = synthetic.peaked_cost_volume(0, 48, 136, 240)
    with torch.no_grad():
        full = torch.nn.functional.interpolate(q.unsqueeze(1), [192, 544, 960], mode="trilinear", align_corners=True).squeeze(1)
        assert maxdiff(full[crows], g["cost_rows"]) <= 1e-6
        fast, plain = O.faster_soft_argmin(full, 192), O.soft_argmin(full, 192)
    assert fast.min().item() < 8.0 and fast.max().item() > 182.0 and float(full.max() - full.min()) > 20.0
    assert maxdiff(fast[sub], g["faster"]) <= 2e-5 and maxdiff(plain[sub], g["plain"]) <= 2e-5


def test_oracle_correlation1d_cost_is_the_samplers_published_semantics():
    """UNPINNED (the sampler package is absent from the reference tree): the oracle against a brute-force statement of
    SpatialCorrelationSampler(kernel_size=1, patch_size=(1, 2D-1), stride=1, padding=0, dilation_patch=1) followed by the
    slicing and activation of correlation1d_cost.py:19-25, and the call-site facts that do not need the package."""
    B, C, H, W, D = 1, 3, 2, 7, 4
    L, R = rand((B, C, H, W), 5), rand((B, C, H, W), 6)
    full = torch.zeros(B, 1, 2 * D - 1, H, W)
    for pw in range(2 * D - 1):
        for x in range(W):
            x2 = x + pw - (D - 1)
            if 0 <= x2 < W:
                full[:, 0, pw, :, x] = (L[:, :, :, x] * R[:, :, :, x2]).sum(1)
    want = torch.nn.functional.leaky_relu(full.squeeze(1)[:, :D], negative_slope=0.1)
    got = O.correlation1d_cost(L, R, D)
    assert got.shape == (B, D, H, W) and torch.allclose(got, want, atol=1e-6)
    assert (got[:, :, :, 0] == torch.nn.functional.leaky_relu(torch.cat([torch.zeros(B, D - 1, H), (L[..., 0] * R[..., 0]).sum(1, keepdim=True)], 1), 0.1)).all()


def test_oracle_fast_volume_gradients_match_the_reference_autograd():
    """The gradients of the sample-based builders: the oracle's differentiable restatement of the sampler against
    gradients of the REFERENCE's fast_cat_fms / fast_dif_fms under torch.autograd (oracle/gen_golden_fast_grad.py), per-pixel
    samples and the builders' own linspace samples."""
    g = golden("fast_volumes_grad.npz")
    for i, row in enumerate(g["cases"]):
        sh, D, seed = tuple(int(v) for v in row[:4]), int(row[4]), int(row[5])
        a, b = rand(sh, seed), rand(sh, seed + 1000)
        gen = torch.Generator().manual_seed(seed + 2000)
        ds = torch.rand((sh[0], D, sh[2], sh[3]), generator=gen) * sh[3] * 0.6 - 2.0
        for kind in ("cat", "dif"):
            ch = 2 * sh[1] if kind == "cat" else sh[1]
            for mode in ("pixel", "default"):
                nd = D if mode == "pixel" else 12
                up = rand((sh[0], ch, nd, sh[2], sh[3]), seed + 3000 + (0 if kind == "cat" else 1))
                kw = dict(disp_sample=ds) if mode == "pixel" else dict(max_disp=24, start_disp=-3, dilation=2)
                dL, dR = O.fast_volume_grads(a, b, up, kind=kind, **kw)
                assert maxdiff(dL, g["%s_%s_dL_%d" % (kind, mode, i)]) <= 1e-5
                assert maxdiff(dR, g["%s_%s_dR_%d" % (kind, mode, i)]) <= 1e-5


FAST_SAMPLE_GRAD_FORMS = [("cat", "cat", {}), ("dif", "dif", {}), ("difn1", "dif", dict(normalize=True, p=1.0)),
                          ("difn2", "dif", dict(normalize=True, p=2.0)), ("difn3", "dif", dict(normalize=True, p=3.0)),
                          ("difnh", "dif", dict(normalize=True, p=0.5))]


def test_oracle_fast_volume_sample_gradients_match_the_reference_autograd():
    """Per-pixel samples that require a gradient (AnyNet.py:60-73, DeepPruner.py:192): d disp_sample, d reference_fm, d target_fm
    of the REFERENCE's builders under torch.autograd, with and without fast_dif_fms's p-norm (dif_fms.py:82-84)."""
    g = golden("fast_volumes_grad.npz")
    for i, row in enumerate(g["cases"]):
        sh, D, seed = tuple(int(v) for v in row[:4]), int(row[4]), int(row[5])
        a, b = rand(sh, seed), rand(sh, seed + 1000)
        gen = torch.Generator().manual_seed(seed + 2000)
        ds = torch.rand((sh[0], D, sh[2], sh[3]), generator=gen) * sh[3] * 0.6 - 2.0
        for name, kind, kw in FAST_SAMPLE_GRAD_FORMS:
            shape = (sh[0], D, sh[2], sh[3]) if kw else (sh[0], (2 if kind == "cat" else 1) * sh[1], D, sh[2], sh[3])
            up = rand(shape, seed + 3000 + (0 if name == "cat" else 1))
            dL, dR, dS = O.fast_volume_grads(a, b, up, kind=kind, disp_sample=ds, wrt_samples=True, **kw)
            assert dS.shape == ds.shape
            for got, key in ((dL, "dL"), (dR, "dR"), (dS, "dS")):
                want = g["%s_samples_%s_%d" % (name, key, i)]
                assert maxdiff(got, want) <= 1e-5 * max(1.0, float(np.abs(want).max())), (name, key, i)
            if kw:
                assert maxdiff(O.fast_dif_fms(a, b, disp_sample=ds, **kw), g["%s_samples_out_%d" % (name, i)]) <= 1e-5


def test_oracle_gwc_has_two_witnesses():
    """Second witness for the UNPINNED group-wise correlation volume: two statements that share no code pin
    each other, and one of them is pinned to the reference.
      (i) G = C (one channel per group): the volume is the element-wise product of the two halves of cat_fms's volume --
          and cat_fms IS pinned bit for bit to the reference (volumes.npz): same shifting convention, same zero region;
     (ii) G = 1, rescaled by C: the per-pixel dot product over all channels = correlation1d_cost's channels in disparity order
          (channel j = disparity D-1-j, correlation1d_cost.py:12-25) before its leaky-ReLU."""
    B, C, H, W, D = 2, 6, 3, 23, 9
    L, R = rand((B, C, H, W), 31), rand((B, C, H, W), 32)
    cat = O.cat_fms(L, R, D, 0, 1)
    assert torch.equal(O.gwc_fms(L, R, D, 0, 1, C), cat[:, :C] * cat[:, C:])           # (i): bit-exact
    dot = O.gwc_fms(L, R, D, 0, 1, 1)[:, 0] * C                                          # [B, D, H, W], disparity d = plane d
    cor = O.correlation1d_cost(L, R, D)                                                  # channel j = disparity D-1-j
    assert (torch.nn.functional.leaky_relu(dot, 0.1) - cor.flip(1)).abs().max().item() <= 1e-5
    # with a start offset and a dilation the group-wise volume still follows cat_fms's index list (cat_fms.py:26-44)
    cat = O.cat_fms(L, R, 12, -3, 2)
    assert torch.equal(O.gwc_fms(L, R, 12, -3, 2, C), cat[:, :C] * cat[:, C:])


@pytest.mark.parametrize("kind", ["cat", "dif"])
def test_first_layer_from_maps_is_the_3d_convolution_of_the_volume(kind):
    """The algebra csrc/catconv.hip rests on, on CPU in FP64: the aggregator's first convolution applied to cat_fms's (or
    dif_fms's) volume equals the sum of 2-D maps of the two feature maps -- every border included (z = 0 / D-1, the columns
    next to x == z, x == 0, x == W-1, the image's first and last rows)."""
    B, C, H, W, D, Co = 2, 4, 7, 19, 6, 5
    L, R = rand((B, C, H, W), 11).double(), rand((B, C, H, W), 12).double()
    w = rand((Co, 2 * C if kind == "cat" else C, 3, 3, 3), 13).double()
    vol = O.cat_fms(L.float(), R.float(), D, 0, 1).double()          # copies: exact
    if kind == "dif":                                                 # L - R in FP64 (the oracle's dif_fms rounds it to FP32)
        vol = vol[:, :C] - vol[:, C:]
        assert (vol.float() - O.dif_fms(L.float(), R.float(), D, 0, 1)).abs().max().item() <= 1e-6
    want = torch.nn.functional.conv3d(vol, w, padding=1)
    got = O.first_layer_from_maps(L, R, w, D, kind)
    assert got.shape == want.shape and (got - want).abs().max().item() <= 1e-12


def _spn_literal(X, G1, G2, G3, horizontal, reverse):
    """Per-element statement of the scan from the kernel source's index arithmetic (gaterecurrent2dnoind_kernel.cu: gate lookup
    :10-98, one line per direction :130-286, line order :535-600): the second witness of the oracle's vectorised restatement."""
    N, C, H, W = X.shape
    out = torch.zeros_like(X)

    def get(d, h, w):
        return d[h, w] if 0 <= h < H and 0 <= w < W else 0.0

    def gate(G, h1, w1, h2, w2):
        if not (0 <= h1 < H and 0 <= w1 < W and 0 <= h2 < H and 0 <= w2 < W):
            return 0.0
        if horizontal:
            first = (w1 > w2) if not reverse else (w1 < w2)
        else:
            first = (h1 > h2) if not reverse else (h1 < h2)
        return G[h1, w1] if first else G[h2, w2]

    for n in range(N):
        for c in range(C):
            x, g1, g2, g3, o = X[n, c], G1[n, c], G2[n, c], G3[n, c], out[n, c]
            S = W if horizontal else H
            for s in (range(S - 1, -1, -1) if reverse else range(S)):
                p = s + 1 if reverse else s - 1
                for t in range(H if horizontal else W):
                    h, w = (t, s) if horizontal else (s, t)
                    nb = [(t - 1, p), (t, p), (t + 1, p)] if horizontal else [(p, t - 1), (p, t), (p, t + 1)]
                    a, b, cc = (gate(g, h, w, nh, nw) for g, (nh, nw) in zip((g1, g2, g3), nb))
                    o[h, w] = (1 - a - b - cc) * x[h, w] + a * get(o, *nb[0]) + b * get(o, *nb[1]) + cc * get(o, *nb[2])
    return out


@pytest.mark.parametrize("horizontal", [True, False])
@pytest.mark.parametrize("reverse", [False, True])
def test_oracle_spn_scan_has_a_literal_witness(horizontal, reverse):
    """The spatial propagation scan (dmb/ops/spn, the reference's only native op -- CUDA, cannot be run here: UNPINNED): the
    oracle's line-at-a-time restatement against a per-element loop written from the kernel source's index arithmetic, all four
    scan directions; and the properties the recurrence implies: the first scanned line is a copy of X, zero gates give H = X."""
    g = torch.Generator().manual_seed(3)
    X = torch.randn(1, 2, 5, 7, generator=g, dtype=torch.float64)
    Gs = [torch.rand(1, 2, 5, 7, generator=g, dtype=torch.float64) * 0.33 for _ in range(3)]
    got = O.spn_gaterecurrent2d(X, *Gs, horizontal, reverse)
    assert (got - _spn_literal(X, *Gs, horizontal, reverse)).abs().max().item() <= 1e-14
    first = (slice(None), slice(None), slice(None), -1 if reverse else 0) if horizontal else (slice(None), slice(None), -1 if reverse else 0)
    assert torch.equal(got[first], X[first])
    zero = [torch.zeros_like(X)] * 3
    assert torch.equal(O.spn_gaterecurrent2d(X, *zero, horizontal, reverse), X)


# ------------------------------------------------------------------------------------------------- the relaxed parity contract
def _fma_chain(p):
    """acc = fmaf(p[:, k], k, acc) for k ascending in FP32, vectorised over pixels: the product p * k is exact in extended precision
    (24 x 8 bits) and the sum carries 64 mantissa bits before the single rounding to FP32 (x87 long double), so each step is a
    fused multiply-add up to a double-rounding event of probability 2^-40."""
    LD = np.longdouble
    assert np.finfo(LD).nmant >= 63, "needs the x87 80-bit long double"
    acc = np.zeros(p.shape[0], dtype=np.float32)
    for k in range(p.shape[1]):
        acc = (acc.astype(LD) + p[:, k].astype(LD) * LD(k)).astype(np.float32)
    return acc


def _reference_order_regression(cost):
    """What the reference's FasterSoftArgmin (disp_predictors/faster_soft_argmin.py:46-71) computes, as an ORDER: torch's FP32
    softmax over the disparity axis, then the (D, 1, 1) convolution with the sample values = a k-ascending FP32 FMA chain."""
    D = cost.shape[1]
    return _fma_chain(torch.softmax(cost, 1).permute(0, 2, 3, 1).reshape(-1, D).numpy())


def _exact_regression(cost):
    c = cost.double()
    k = torch.arange(c.shape[1], dtype=torch.float64).view(1, -1, 1, 1)
    return (torch.softmax(c, 1) * k).sum(1).reshape(-1).numpy()


def test_reference_faster_soft_argmin_is_an_fp32_fma_chain_bit_for_bit():
    """The proof behind bench.py's PARITY_CONTRACT, part 1: the reference's FasterSoftArgmin output -- recorded from the reference
    itself in predictors.npz (D = 192, 120 pixels) and fullsize_regression_ends.npz (8160 pixels of the 544x960 map, disparities
    5 .. 185) -- equals softmax + k-ascending FP32 FMA chain BIT FOR BIT.  So the reference's rounding is reproducible as an order."""
    from densematchingbenchmark_amd import synthetic
    g = golden("predictors.npz")
    D, seed = (int(v) for v in g["d192_meta"])
    cost = rand((2, D, 6, 10), seed, float(g["d192_gain"][0]))
    assert np.array_equal(_reference_order_regression(cost).reshape(2, 1, 6, 10), g["d192_faster"])
    g2 = golden("fullsize_regression_ends.npz")
    q = synthetic.peaked_cost_volume(0, 48, 136, 240)
    with torch.no_grad():
        full = torch.nn.functional.interpolate(q.unsqueeze(1), [192, 544, 960], mode="trilinear", align_corners=True).squeeze(1)
        p = torch.softmax(full, 1)[:, :, 3::8, 5::8].permute(0, 2, 3, 1).reshape(-1, 192).numpy()
    assert np.array_equal(_fma_chain(p).reshape(1, 1, 68, 120), g2["faster"])


def test_reproducing_the_reference_order_cannot_meet_1e4_without_bit_identical_costs():
    """Part 2 (why the contract is max(1e-4, 1.25 x the reference arithmetic's own distance from FP64) and the kernel accumulates in
    FP64).  On a seeded D = 192 volume of 32 768 pixels, with the reference's regression restated as the order part 1 pins:
      (a) the reference's OWN output is already > 1e-4 from the exact value;
      (b) costs moved by at most 3e-6 (the distance between two correct FP32 evaluations of the aggregator) move
          the exact value by < 4e-5, yet the reference-order chain on them lands > 1e-4 from the reference's output on the
          unperturbed costs: its 192 roundings at magnitude ~100 (ulp 7.6e-6) decorrelate under ANY input perturbation, so only
          bit-identical costs reproduce them -- while the FP64-accumulated regression of the perturbed costs stays as close to the
          reference's output as the reference is to the truth;
      (c) at the cost scale of the peaked fixtures (+-30) two volumes that differ by ONE ulp per element have exact disparities
          > 2e-4 apart: at max_disp 192 the 1e-4 bound is below the conditioning of the regression itself."""
    flat = rand((1, 192, 128, 256), 77, 0.3)
    ref_out, exact = _reference_order_regression(flat), _exact_regression(flat)
    cost1 = rand((1, 192, 128, 256), 77, 1.0)
    assert np.abs(_reference_order_regression(cost1) - _exact_regression(cost1)).max() > 1.0e-4                     # (a)
    gen = torch.Generator().manual_seed(5)
    moved = flat + (torch.rand(flat.shape, generator=gen) * 2 - 1) * 3e-6
    exact_moved = _exact_regression(moved)
    assert np.abs(exact_moved - exact).max() < 4e-5                                                                   # (b) the truth barely moves
    assert np.abs(_reference_order_regression(moved) - ref_out).max() > 1.0e-4                                        # ... the chain does not follow
    assert np.abs(exact_moved - ref_out).max() <= np.abs(exact - ref_out).max() + 4e-5                                # ... FP64 accumulation does
    peaked = rand((1, 192, 128, 256), 77, 8.0)
    gen = torch.Generator().manual_seed(5)
    sgn = (torch.randint(0, 3, peaked.shape, generator=gen) - 1).double()
    neighbour = (peaked.double() * (1 + sgn * 2.0 ** -23)).float()
    assert (neighbour - peaked).abs().max().item() < 4e-6
    assert np.abs(_exact_regression(neighbour) - _exact_regression(peaked)).max() > 2e-4                              # (c)


# ------------------------------------------------------------------------------------------------- real data, headline size
def _demo_model_params():
    import os
    from densematchingbenchmark_amd import synthetic
    from densematchingbenchmark_amd.config import Config
    from densematchingbenchmark_amd.modeling import build_model
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    model = build_model(Config.fromfile(os.path.join(root, "configs", "PSMNet", "scene_flow.py"))).eval()   # backbone included
    synthetic.init_params_(model, seed=21, classif_gain=10.0)
    return {k: v.clone() for k, v in model.state_dict().items()}


def _demo_files(g, tmp_path):
    paths = {}
    for key, name in (("left_image_path", "left.png"), ("right_image_path", "right.png"), ("left_disp_map_path", "left.pfm")):
        p = tmp_path / name
        p.write_bytes(g["file_" + key].tobytes())
        paths[key] = str(p)
    return paths


def test_oracle_data_conventions_and_whole_model_on_the_reference_demo_pair(tmp_path):
    """tests/golden/demo_sceneflow.npz (oracle/gen_golden_demo.py: the reference's own inference_stereo on its 540x960 demo pair
    with pad_to_shape = (544, 960)): the oracle's imread -> [:, :, :3] -> StereoPad -> Normalize equals the tensor the reference's
    transforms fed its model BIT FOR BIT (padding rows hold -mean / std), and the oracle's whole PSMNet (backbone + path) on it
    reproduces the reference's three disparity maps, the cost rows and the per-image error dicts."""
    from densematchingbenchmark_amd.data import imread
    from densematchingbenchmark_amd.disp_io import load_scene_flow_disp
    g = golden("demo_sceneflow.npz")
    paths = _demo_files(g, tmp_path)
    li, ri = (O.prepare_image(imread(paths[k]), pad_to_shape=(544, 960)) for k in ("left_image_path", "right_image_path"))
    assert tuple(li.shape) == tuple(g["padded_shape"]) == (1, 3, 544, 960)
    assert np.array_equal(li[:, :, ::17, :].numpy(), g["left_rows"]) and np.array_equal(ri[:, :, 3::31, :].numpy(), g["right_rows"])
    assert li.double().sum().item() == g["left_sum_f64"][0] and ri.double().abs().sum().item() == g["right_sum_f64"][1]
    pad = torch.tensor([-m / s for m, s in zip(O.IMAGENET_MEAN, O.IMAGENET_STD)])
    assert maxdiff(li[0, :, 0, 0], pad) <= 1e-6 and maxdiff(li[0, :, 3, 959], pad) <= 1e-6      # padded BEFORE normalisation
    gt = torch.from_numpy(np.ascontiguousarray(load_scene_flow_disp(paths["left_disp_map_path"])))
    assert np.array_equal(gt[::45].numpy(), g["ori_left_disp_rows"])
    # the crop branch (the reference's CenterCrop + Normalize on the same files): images and disparity
    crop = O.normalize(O.center_crop(O.image_to_chw(imread(paths["left_image_path"])), (512, 896)))
    assert tuple(crop.shape) == tuple(g["crop_shape"]) and np.array_equal(crop[:, ::37, :].numpy(), g["crop_left_rows"])
    assert np.array_equal(O.center_crop(gt[None], (512, 896))[:, ::37, :].numpy(), g["crop_disp_rows"])
    p = _demo_model_params()
    with torch.no_grad():
        disps, costs = O.psmnet_model(li, ri, p, 192)
    for i, (d, c) in enumerate(zip(disps, costs)):
        d, c = O.remove_padding(d, (540, 960)), O.remove_padding(c, (540, 960))
        assert tuple(d.shape) == tuple(g["cropped_shape"]) and tuple(c.shape) == tuple(g["cost_shape"])
        assert maxdiff(d[:, :, ::4, ::4], g["disp%d_s4" % i]) <= 3e-5
        assert maxdiff(c[:, ::24, 5::107, :], g["cost%d_rows" % i]) <= 3e-5
        err = O.calc_error(d, gt[None, None], 0, 192)
        want = dict(zip(("epe", "1px", "2px", "3px", "5px"), g["err%d" % i]))
        assert abs(err["epe"] - want["epe"]) <= 1e-5 and all(abs(err[k] - want[k]) <= 1e-3 for k in ("1px", "2px", "3px", "5px"))
        if i == 0:
            assert maxdiff(d, g["disp0_full"]) <= 3e-5


def test_reference_self_spread_fixture_is_consistent_with_the_other_fullsize_fixtures():
    """Round 6: fullsize_psmnet_spread.npz (oracle/gen_golden_fullsize.py `spread`: the REAL reference at 1 / 3 / 8 host threads) --
    its 8-thread maps ARE the maps of the older full-size fixtures bit for bit (same reference, same weights, inputs and thread
    count), the stored scalars are the maxima of the stored maps' differences, and the self-spread is of the order of north_star's
    bound itself: what makes `1.6 x spread` a principled bound on |hip - reference| (tests/test_fullsize_gpu.py)."""
    g = golden("fullsize_psmnet_spread.npz")
    for tag, other, key in (("s544", "fullsize_psmnet.npz", "pair0_disp%d"), ("kitti", "fullsize_psmnet_kitti.npz", "pair0_disp%d"),
                            ("g30", "fullsize_psmnet_gain30.npz", "pair0_disp%d")):
        o = golden(other)
        for k in (1, 2, 3):
            assert np.array_equal(g["%s_t8_disp%d" % (tag, k)], o[key % k]), (tag, k)
            sub = max(np.abs(g["%s_t%d_disp%d" % (tag, a, k)] - g["%s_t%d_disp%d" % (tag, b, k)]).max() for a, b in ((8, 3), (8, 1), (3, 1)))
            assert abs(float(g["%s_spread_sub_disp%d" % (tag, k)]) - float(sub)) <= 1e-12
            full = float(g["%s_spread_full_disp%d" % (tag, k)])
            assert full >= float(sub) and 3e-5 <= full <= 3e-4, (tag, k, full)
    for tag in ("s544", "kitti"):
        d = max(np.abs(g["%s_t%d_minus_t8_disp3_full" % (tag, t)]).max() for t in (1, 3))
        assert d <= float(g["%s_spread_full_disp3" % tag]) + 1e-12


def test_oracle_on_the_peaked_trained_weights_fixture():
    """Round 6: the oracle's FP32 path on the trained-weights fixture (tests/golden/psmnet_trained_weights.npz + the banded exact-match
    pair) against what the REAL reference returned there (fullsize_psmnet_peaked.npz): the same arithmetic on the same library, so
    agreement to the thread-count noise; the fixture is peaked (E|k - disp| 1.8 px, EPE 1.2 px) and its own record of the reference's
    self-spread is consistent."""
    from densematchingbenchmark_amd import synthetic
    sub = (slice(None), slice(None), slice(3, None, 8), slice(5, None, 8))
    crows = (slice(None), slice(7, None, 48), slice(11, None, 136), slice(None))
    g, w = golden("fullsize_psmnet_peaked.npz"), golden("psmnet_trained_weights.npz")
    p = {k: torch.from_numpy(w[k].astype("float32") if w[k].dtype.kind == "f" else w[k]) for k in w.files}
    lf, rf, gt = synthetic.banded_match_pair(7, 136, 240, 48, bands=6)
    with torch.no_grad():
        disps, costs = O.psmnet_path(lf, rf, p, 192)
    for lvl in range(3):
        assert maxdiff(disps[lvl][sub], g["disp%d" % (3 - lvl)]) <= 1.6 * float(g["spread_full_disp%d" % (3 - lvl)])
        assert maxdiff(costs[lvl][crows], g["cost%d_rows" % (3 - lvl)]) <= 2e-4          # costs up to 65: a few ulp
    assert maxdiff(disps[0], g["disp3_full"]) <= 1.6 * float(g["spread_full_disp3"])
    width = g["disp3_width_full"].astype("float32")
    assert float((width < 2.0).mean()) >= 0.9 and 1.0 <= float(g["stats"][0]) <= 2.5 and float(g["stats"][2]) <= 1.5
    assert abs(float(g["disp3_self_spread_full"].max()) - float(g["spread_full_disp3"])) <= 1e-12
    mask = gt > 0
    assert abs((disps[0][mask] - gt[mask]).abs().mean().item() - float(g["stats"][2])) <= 1e-5
