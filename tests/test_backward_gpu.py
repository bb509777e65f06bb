"""Backward passes of the convolution units against the CPU oracle's autograd.

Tolerances: the kernels are FP32 fma chains in a different summation order than the oracle's FP32 CPU evaluation, so
values are compared against an FP64 evaluation with the FP32 oracle's own distance to FP64 as the yardstick."""
import pytest
import torch

from oracle import dmb_oracle as O

pytestmark = pytest.mark.gpu


def _ops():
    from densematchingbenchmark_amd import ops

    return ops


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _close(got, ref64, ref32, what, floor=2e-6):
    """|got - FP64| must stay within 4x the FP32 oracle's own error (plus a floor, default 2e-6, of the value range)."""
    scale = ref64.abs().max().item()
    err = (got.double() - ref64).abs().max().item()
    own = (ref32.double() - ref64).abs().max().item()
    assert err <= 4 * own + floor * scale, "%s: error %.3e vs oracle FP32 error %.3e (range %.3e)" % (what, err, own, scale)


@pytest.mark.parametrize("Ci,Co,shape", [
    (32, 32, (1, 5, 8, 24)),
    (32, 32, (2, 9, 7, 28)),        # partial tiles in y and x, two z segments
    (64, 32, (1, 4, 9, 48)),
    (32, 64, (1, 6, 6, 24)),
    (8, 40, (1, 3, 5, 12)),         # channel counts that are not multiples of 32
    (32, 32, (1, 4, 6, 30)),        # W % 4 != 0 -> dword staging
    (32, 32, (1, 20, 12, 48)),
    (32, 32, (1, 6, 10, 128)),      # 32-column tiles
    (64, 32, (2, 5, 6, 64)),
    (32, 1, (2, 6, 9, 70)),         # classifier head: the single-output-channel kernel
    (6, 1, (1, 3, 4, 5)),
])
def test_conv3d_wgrad(dev, Ci, Co, shape):
    ops = _ops()
    B, D, H, W = shape
    x, dc = _rand((B, Ci, D, H, W), 1), _rand((B, Co, D, H, W), 2)
    w = _rand((Co, Ci, 3, 3, 3), 3, 0.05)
    _, dw32 = O.conv3d_backward(x, w, dc)
    _, dw64 = O.conv3d_backward(x, w, dc, dtype=torch.float64)
    got = ops.conv3d_k3_wgrad(x.to(dev), dc.to(dev)).cpu()
    assert got.shape == (Co, Ci, 3, 3, 3)
    _close(got, dw64, dw32, "dW")
    again = ops.conv3d_k3_wgrad(x.to(dev), dc.to(dev)).cpu()
    assert torch.equal(got, again), "the weight gradient must be bit-reproducible (no atomics)"


@pytest.mark.parametrize("Ci,Co,stride,shape", [
    (32, 32, 1, (1, 5, 8, 48)),
    (64, 32, 1, (1, 4, 9, 30)),
    (32, 64, 1, (2, 4, 8, 24)),
    (32, 64, 2, (1, 8, 12, 40)),
    (64, 64, 2, (1, 4, 8, 24)),
    (32, 64, 2, (1, 7, 9, 21)),     # odd extents: the even-size adjoint with its last plane / row / column dropped
])
def test_conv3d_dgrad(dev, Ci, Co, stride, shape):
    ops = _ops()
    B, D, H, W = shape
    x = _rand((B, Ci, D, H, W), 1)
    w = _rand((Co, Ci, 3, 3, 3), 3, 0.05)
    dc = _rand((B, Co, (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1), 2)
    dx32, _ = O.conv3d_backward(x, w, dc, stride)
    dx64, _ = O.conv3d_backward(x, w, dc, stride, dtype=torch.float64)
    got = ops.conv3d_k3_dgrad(dc.to(dev), w.to(dev), stride, (D, H, W)).cpu()
    assert got.shape == x.shape
    _close(got, dx64, dx32, "dx")


@pytest.mark.parametrize("Ci,Co,shape", [(64, 64, (1, 3, 4, 12)), (64, 32, (1, 4, 6, 20))])
def test_deconv3d_dgrad(dev, Ci, Co, shape):
    ops = _ops()
    B, D, H, W = shape
    x = _rand((B, Ci, D, H, W), 1)
    w = _rand((Ci, Co, 3, 3, 3), 3, 0.05)
    dy = _rand((B, Co, 2 * D, 2 * H, 2 * W), 2)
    dx32, _ = O.deconv3d_backward(x, w, dy)
    dx64, _ = O.deconv3d_backward(x, w, dy, dtype=torch.float64)
    got = ops.deconv3d_k3s2_dgrad(dy.to(dev), w.to(dev)).cpu()
    assert got.shape == x.shape
    _close(got, dx64, dx32, "dx")


@pytest.mark.parametrize("shape", [(2, 8, 5, 6, 12), (3, 32, 4, 7, 9), (2, 5, 33, 20)])   # the last one is 2-D, S % 4 != 0 above
@pytest.mark.parametrize("relu,with_res", [(0, False), (1, False), (1, True), (2, True), (0, True)])
@pytest.mark.parametrize("training", [True, False])
def test_bn_act_train(dev, shape, relu, with_res, training):
    """Batch statistics, running-buffer update, normalising pass and the backward of one BN(+skip)(+ReLU) epilogue."""
    ops = _ops()
    C = shape[1]
    c = _rand(shape, 1) * 1.5 + 0.3
    gamma, beta = _rand((C,), 2) * 0.5 + 1.0, _rand((C,), 3) * 0.2
    res = _rand(shape, 4) if with_res else None
    dy = _rand(shape, 5)
    rm, rv = _rand((C,), 6) * 0.1, _rand((C,), 7).abs() + 0.5
    ref = O.bn_act_train(c, gamma, beta, res, relu, dy=dy, training=training, running_mean=rm, running_var=rv)
    ref64 = O.bn_act_train(c, gamma, beta, res, relu, dy=dy, dtype=torch.float64, training=training, running_mean=rm, running_var=rv)
    mode = {0: False, 1: True, 2: "pre"}[relu]
    cg = c.to(dev)
    if training:
        rmg, rvg = rm.to(dev), rv.to(dev)
        mean, invstd, scale, shift = ops.bn_train_stats(cg, gamma.to(dev), beta.to(dev), rmg, rvg, momentum=0.1, eps=1e-5)
        # tolerance: FP32 statistics of O(1) data over <= 1e4 elements
        assert (rmg.cpu() - ref["running_mean"]).abs().max().item() < 1e-5
        assert (rvg.cpu() - ref["running_var"]).abs().max().item() < 1e-5
    else:
        invstd = (1.0 / torch.sqrt(rv + 1e-5)).to(dev)
        mean = rm.to(dev)
        scale = gamma.to(dev) * invstd
        shift = beta.to(dev) - mean * scale
    y = ops.bn_act(cg, scale, shift, res.to(dev) if with_res else None, relu=mode)
    assert (y.cpu() - ref["y"]).abs().max().item() < 2e-5
    dc, dgamma, dbeta, dres = ops.bn_act_bwd(dy.to(dev), cg, y, scale, shift, mean, invstd, relu=mode, training=training,
                                             want_dres=with_res)
    # elements whose pre-activation is within rounding of zero may take the other ReLU branch: compare away from them
    _close(dgamma.cpu(), ref64["dgamma"], ref["dgamma"], "dgamma")
    _close(dbeta.cpu(), ref64["dbeta"], ref["dbeta"], "dbeta")
    _close(dc.cpu(), ref64["dc"], ref["dc"], "dc")
    if with_res:
        assert torch.equal(dres.cpu(), ref["dres"])


@pytest.mark.parametrize("shape,md,sd,dil", [
    ((2, 8, 6, 40), 12, 0, 1),
    ((1, 5, 7, 61), 12, -3, 1),
    ((1, 4, 5, 12), 20, 0, 1),      # disparities beyond the width
    ((1, 8, 9, 64), 6, 0, 2),
])
def test_volume_builders_backward(dev, shape, md, sd, dil):
    """cat_fms / dif_fms: the adjoint sums; compared with autograd through the oracle's builders.  Tolerance: an FP32 sum
    of <= 20 terms in a different order (1e-5 on O(1) data)."""
    ops = _ops()
    B, C, H, W = shape
    L, R = _rand(shape, 1), _rand(shape, 2)
    idx = ops.disp_index_list(md, sd, dil)
    for name, builder, bwd, VC in (("cat", O.cat_fms, ops.cat_fms_bwd, 2 * C), ("dif", O.dif_fms, ops.dif_fms_bwd, C)):
        dvol = _rand((B, VC, len(idx), H, W), 3)
        Lr, Rr = L.clone().requires_grad_(True), R.clone().requires_grad_(True)
        # the oracle writes slices of leaf-derived tensors into a zero volume: differentiable as is
        vol = builder(Lr, Rr, md, sd, dil)
        dLr, dRr = torch.autograd.grad(vol, (Lr, Rr), dvol)
        dL, dR = bwd(dvol.to(dev), idx)
        assert (dL.cpu() - dLr).abs().max().item() < 1e-5, name
        assert (dR.cpu() - dRr).abs().max().item() < 1e-5, name


@pytest.mark.parametrize("B,D,H,W,alpha", [(2, 24, 5, 17, 1.0), (1, 192, 3, 8, 1.0), (1, 16, 4, 4, -1.5)])
def test_soft_argmin_backward(dev, B, D, H, W, alpha):
    ops = _ops()
    cost = _rand((B, D, H, W), 1, 2.0)
    g = _rand((B, 1, H, W), 2)
    vals = O.disp_sample_values(D).tolist()
    c = cost.clone().double().requires_grad_(True)
    p = torch.softmax(c * alpha, dim=1)
    disp64 = (p * torch.tensor(vals, dtype=torch.float64).view(1, D, 1, 1)).sum(1, keepdim=True)
    ref, = torch.autograd.grad(disp64, c, g.double())
    disp = ops.soft_argmin(cost.to(dev), vals, alpha, True)
    got = ops.soft_argmin_bwd(cost.to(dev), disp, g.to(dev), vals, alpha).cpu()
    # tolerance: __expf and an FP32 quotient against FP64, relative to the largest gradient entry
    assert (got.double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("lo,scale", [((6, 5, 9), 4), ((4, 7, 12), 4), ((8, 4, 6), 2), ((5, 3, 4), 1),
                                      # (round 6) the row-group form of the (y, x) contraction: rows not a multiple of its groups of 8,
                                      # more than 256 columns (two column passes), a width that does not divide 256, two row lanes
                                      ((3, 9, 40), 4), ((2, 5, 300), 2), ((2, 17, 24), 3), ((4, 20, 96), 4),
                                      # ... and a window of more than 16 columns (6x): the one-thread-per-voxel form
                                      ((2, 4, 6), 6)])
def test_upsample_regression_backward(dev, lo, scale):
    """d disp / d low-resolution cost through F.interpolate(trilinear, align_corners=True) + softmax regression."""
    import torch.nn.functional as F
    ops = _ops()
    Di, Hi, Wi = lo
    Do, Ho, Wo = Di * scale, Hi * scale, Wi * scale
    x = _rand((2, Di, Hi, Wi), 1, 2.0)
    g = _rand((2, 1, Ho, Wo), 2)
    vals = O.disp_sample_values(Do).tolist()
    xr = x.clone().double().requires_grad_(True)
    up = F.interpolate(xr.unsqueeze(1), size=(Do, Ho, Wo), mode="trilinear", align_corners=True).squeeze(1)
    p = torch.softmax(up, dim=1)
    disp64 = (p * torch.tensor(vals, dtype=torch.float64).view(1, Do, 1, 1)).sum(1, keepdim=True)
    ref, = torch.autograd.grad(disp64, xr, g.double(), retain_graph=True)
    _, disp = ops.trilinear_ac_soft_argmin(x.to(dev), (Do, Ho, Wo), vals, 1.0)
    got = ops.trilinear_ac_soft_argmin_bwd(x.to(dev), disp, g.to(dev), (Do, Ho, Wo), vals, 1.0).cpu()
    assert got.shape == x.shape
    # tolerance: FP32 interpolation weights + __expf against an FP64 evaluation, relative to the largest entry; an FP32 source
    # coordinate carries an absolute error of one ulp of its magnitude (3e-5 at column 600), and the weights with it: wide maps get
    # the bound of a 100-column one times Wo / 100 (the reference's FP32 F.interpolate has the same weights)
    wide = max(1.0, Wo / 100.0)
    assert (got.double() - ref).abs().max().item() <= 5e-5 * wide * max(1.0, ref.abs().max().item())
    # a gradient on the up-sampled volume itself, alone and together with the disparity's
    gv = _rand((2, Do, Ho, Wo), 3)
    refv, = torch.autograd.grad(up, xr, gv.double(), retain_graph=True)
    gotv = ops.trilinear_ac_bwd(gv.to(dev), (Di, Hi, Wi)).cpu()
    assert (gotv.double() - refv).abs().max().item() <= 1e-5 * wide * max(1.0, refv.abs().max().item())
    both = ops.trilinear_ac_soft_argmin_bwd(x.to(dev), disp, g.to(dev), (Do, Ho, Wo), vals, 1.0, grad_cost=gv.to(dev)).cpu()
    assert (both.double() - (ref + refv)).abs().max().item() <= 5e-5 * wide * max(1.0, (ref + refv).abs().max().item())


@pytest.mark.parametrize("Ci,Co,shape", [
    (32, 64, (1, 8, 12, 24)),
    (64, 64, (2, 6, 8, 48)),
    (32, 32, (1, 7, 9, 40)),        # odd extents of the big tensor (2n - 1)
    (16, 40, (1, 4, 4, 8)),
    (32, 64, (1, 24, 20, 56)),
    (32, 32, (1, 4, 6, 12)),        # widths 12 -> 6: dword staging
    (32, 32, (2, 5, 7, 27)),        # odd everything
])
def test_conv3d_s2_wgrad(dev, Ci, Co, shape):
    ops = _ops()
    B, D, H, W = shape
    x = _rand((B, Ci, D, H, W), 1)
    w = _rand((Co, Ci, 3, 3, 3), 3, 0.05)
    dc = _rand((B, Co, (D - 1) // 2 + 1, (H - 1) // 2 + 1, (W - 1) // 2 + 1), 2)
    _, dw32 = O.conv3d_backward(x, w, dc, 2)
    _, dw64 = O.conv3d_backward(x, w, dc, 2, dtype=torch.float64)
    got = ops.conv3d_k3s2_wgrad(x.to(dev), dc.to(dev)).cpu()
    _close(got, dw64, dw32, "dW (stride 2)")


@pytest.mark.parametrize("Ci,Co,shape", [(64, 64, (1, 3, 4, 12)), (64, 32, (2, 4, 6, 20)), (32, 8, (1, 5, 5, 8))])
def test_deconv3d_wgrad(dev, Ci, Co, shape):
    ops = _ops()
    B, D, H, W = shape
    x = _rand((B, Ci, D, H, W), 1)
    w = _rand((Ci, Co, 3, 3, 3), 3, 0.05)
    dy = _rand((B, Co, 2 * D, 2 * H, 2 * W), 2)
    _, dw32 = O.deconv3d_backward(x, w, dy)
    _, dw64 = O.deconv3d_backward(x, w, dy, dtype=torch.float64)
    got = ops.deconv3d_k3s2_wgrad(x.to(dev), dy.to(dev)).cpu()
    assert got.shape == w.shape
    _close(got, dw64, dw32, "dW (transposed)")


def test_psmnet_training_step(dev):
    """One training iteration of the PSMNet cost path through build_model(cfg, backbone=None) in train() mode: losses, the gradient of
    every parameter and of both feature maps, and the BatchNorm running buffers against the oracle's autograd.

    Tolerances are written at the asserts (losses 1e-4 relative; gradients see the comment there)."""
    import os
    from densematchingbenchmark_amd.config import Config
    from densematchingbenchmark_amd.modeling import build_model
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", "PSMNet", "scene_flow.py"))
    md = 32
    cfg.model.max_disp = md
    cfg.model.cost_processor.cost_computation.max_disp = md // 4
    cfg.model.cost_processor.cost_aggregator.max_disp = md
    cfg.model.disp_predictor.max_disp = md
    cfg.model.losses.l1_loss.max_disp = md
    p = O.random_params_psm(seed=7, classif_gain=4.0)
    model = build_model(cfg, backbone=None)
    sd = {"cost_processor.aggregator." + k: v for k, v in p.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected
    model = model.to(dev).train()
    lf, rf = _rand((2, 32, 8, 24), 11), _rand((2, 32, 8, 24), 12)
    gt = torch.rand((2, 1, 32, 96), generator=torch.Generator().manual_seed(13)) * 40.0 - 4.0   # some pixels outside (0, 32)
    pp = O.with_prefix(p, "cost_processor.aggregator.")
    losses32, grads32, running32 = O.psmnet_train_step(lf, rf, pp, md, gt)
    losses64, grads64, _ = O.psmnet_train_step(lf, rf, pp, md, gt, dtype=torch.float64)

    lfg, rfg = lf.to(dev).requires_grad_(True), rf.to(dev).requires_grad_(True)
    results, loss_dict = model(dict(leftFeature=lfg, rightFeature=rfg, leftDisp=gt.to(dev)))
    assert results == {} and sorted(loss_dict) == ["l1_loss_lvl0", "l1_loss_lvl1", "l1_loss_lvl2"]
    for i in range(3):
        assert abs(loss_dict["l1_loss_lvl%d" % i].item() - losses64[i].item()) <= 1e-4 * max(1.0, abs(losses64[i].item()))
    sum(loss_dict.values()).backward()

    # A ReLU whose pre-activation lies within FP32 rounding of zero may take the other branch here than in the oracle
    # (different summation order): about one element in 1e6, and with the sparse gradients of a peaked soft-argmin one such
    # element moves a whole tensor's gradient by up to ~1e-2 of its range -- everything downstream of it with it.  So:
    # every gradient within 3e-2 of its range (a wrong mask, a missing term or a wrong adjoint is O(1)), and at least 60 %
    # of the tensors within 4x the FP32 oracle's own distance to FP64 (floor 2e-5 of the range).
    named = dict(model.named_parameters())
    tight, checked = 0, 0
    for k, g64 in grads64.items():
        if k in ("ref_fms", "tgt_fms"):
            got = (lfg if k == "ref_fms" else rfg).grad
        else:
            if not named[k].requires_grad:
                continue
            got = named[k].grad
        assert got is not None, k
        scale = g64.abs().max().item()
        err = (got.cpu().double() - g64).abs().max().item()
        own = (grads32[k].double() - g64).abs().max().item()
        assert err <= 3e-2 * scale, "grad of %s: error %.3e of range %.3e" % (k, err, scale)
        tight += err <= 4 * own + 2e-5 * scale
        checked += 1
    assert checked == len(grads64) == 80   # 28 convolution weights, 25 BatchNorm (gamma, beta) pairs, the two feature maps
    assert tight >= 0.6 * checked, "only %d of %d gradients within the tight tolerance" % (tight, checked)
    buffers = dict(model.named_buffers())
    for k, v in running32.items():
        assert (buffers[k].cpu() - v).abs().max().item() <= 1e-5 * max(1.0, v.abs().max().item()), k
    # a second call in eval mode under no_grad still takes the fused inference kernels
    model.eval()
    with torch.no_grad():
        out, _ = model(dict(leftFeature=lf.to(dev), rightFeature=rf.to(dev)))
    assert len(out["disps"]) == 3 and not out["disps"][0].requires_grad


def test_stereonet_cost_path_training(dev):
    """dif volume -> StereoNetAggregator (biased convolutions, batch-statistics BatchNorm, head with bias) -> stand-alone
    soft-argmin -> smooth-L1, in train() mode, against the oracle's autograd: exercises the difference-volume backward, the
    bias gradients and the stand-alone soft-argmin backward.  Tolerances as in test_psmnet_training_step."""
    from densematchingbenchmark_amd.modeling.stereo.cost_processors.aggregators import AGGREGATORS
    from densematchingbenchmark_amd.modeling.stereo.cost_processors.utils.dif_fms import dif_fms
    from densematchingbenchmark_amd.modeling.stereo.disp_predictors import PREDICTORS
    from densematchingbenchmark_amd.modeling.stereo.losses import DispSmoothL1Loss
    from tests._util import golden
    g = golden("aggregators.npz")
    p = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("snp_")}
    D = 12
    agg = AGGREGATORS["StereoNet"](max_disp=D, in_planes=32, batch_norm=True, num=4)
    agg.load_state_dict(p, strict=False)
    agg = agg.to(dev).train()
    pred = PREDICTORS['FASTER'](max_disp=D).to(dev)
    lf, rf = _rand((2, 32, 10, 28), 21), _rand((2, 32, 10, 28), 22)
    gt = torch.rand((2, 1, 10, 28), generator=torch.Generator().manual_seed(23)) * 14.0 - 1.0
    loss32, g32, run32 = O.stereonet_train_step(lf, rf, p, D, gt)
    loss64, g64, _ = O.stereonet_train_step(lf, rf, p, D, gt, dtype=torch.float64)
    lfg, rfg = lf.to(dev).requires_grad_(True), rf.to(dev).requires_grad_(True)
    cost = agg(dif_fms(lfg, rfg, D, 0, 1))[0]
    disp = pred(cost)
    loss = DispSmoothL1Loss(max_disp=D).loss_per_level(disp, gt.to(dev))
    assert abs(loss.item() - loss64.item()) <= 1e-4 * max(1.0, abs(loss64.item()))
    loss.backward()
    named = dict(agg.named_parameters())
    tight = 0
    zero = 1e-6 * max(v.abs().max().item() for v in g64.values())   # for gradients that are exactly zero (see below)
    for k, ref in g64.items():
        got = lfg.grad if k == "ref_fms" else rfg.grad if k == "tgt_fms" else named[k].grad
        assert got is not None, k
        scale = ref.abs().max().item()
        err = (got.cpu().double() - ref).abs().max().item()
        # (+zero: the bias of a convolution in front of a batch-statistics BatchNorm has an exactly zero gradient)
        assert err <= 3e-2 * scale + zero, "grad of %s: error %.3e of range %.3e" % (k, err, scale)
        tight += err <= 4 * (g32[k].double() - ref).abs().max().item() + 2e-5 * scale + zero
    assert len(g64) == 4 * 4 + 2 + 2 and tight >= 0.6 * len(g64)   # 4 x (weight, bias, gamma, beta), head weight + bias, 2 features
    buffers = dict(agg.named_buffers())
    for k, v in run32.items():
        assert (buffers[k].cpu() - v).abs().max().item() <= 1e-5 * max(1.0, v.abs().max().item()), k


@pytest.mark.parametrize("shape", [(2, 3, 5, 9), (1, 6, 4, 33)])
def test_deconv_k8s4_backward(dev, shape):
    """AcfNet's learned up-sampling: dx and dw against autograd of F.conv_transpose3d (FP64 yardstick)."""
    import torch.nn.functional as F
    ops = _ops()
    x, w = _rand(shape, 31), _rand((1, 1, 8, 8, 8), 32, 0.1)
    dy = _rand((shape[0], 4 * shape[1], 4 * shape[2], 4 * shape[3]), 33)
    res = {}
    for dt in (torch.float32, torch.float64):
        xr, wr = x.to(dt).requires_grad_(True), w.to(dt).requires_grad_(True)
        y = F.conv_transpose3d(xr.unsqueeze(1), wr, None, stride=4, padding=2).squeeze(1)
        res[dt] = torch.autograd.grad(y, (xr, wr), dy.to(dt))
    dx, dw = ops.deconv3d_k8s4_c1_bwd(x.to(dev), w.view(8, 8, 8).to(dev), dy.to(dev))
    _close(dx.cpu(), res[torch.float64][0], res[torch.float32][0], "dx")
    _close(dw.cpu().view(1, 1, 8, 8, 8), res[torch.float64][1], res[torch.float32][1], "dw")


def test_acfnet_uniform_training_step(dev):
    """One training iteration of AcfNet with a fixed variance (configs/AcfNet/scene_flow_uniform.py): stereo focal loss on the
    three up-sampled cost volumes + smooth-L1 on the disparities, through the learned k8/s4 up-sampling, the stand-alone
    soft-argmin and the biased convolution units.  Tolerances as in test_psmnet_training_step."""
    import os
    from densematchingbenchmark_amd.config import Config
    from densematchingbenchmark_amd.modeling import build_model
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", "AcfNet", "scene_flow_uniform.py"))
    md = 32
    cfg.model.max_disp = md
    cfg.model.cost_processor.cost_computation.max_disp = md // 4
    cfg.model.cost_processor.cost_aggregator.max_disp = md
    cfg.model.disp_predictor.max_disp = md
    cfg.model.losses.l1_loss.max_disp = md
    cfg.model.losses.focal_loss.max_disp = md
    p = O.with_prefix(O.random_params_psm(seed=3, classif_gain=4.0, acf=True), "cost_processor.aggregator.")
    model = build_model(cfg, backbone=None)
    missing, unexpected = model.load_state_dict(p, strict=False)
    assert not unexpected
    model = model.to(dev).train()
    lf, rf = _rand((2, 32, 8, 24), 41), _rand((2, 32, 8, 24), 42)
    gt = torch.rand((2, 1, 32, 96), generator=torch.Generator().manual_seed(43)) * 40.0 - 4.0
    l32, g32, run32 = O.acfnet_uniform_train_step(lf, rf, p, md, gt)
    l64, g64, _ = O.acfnet_uniform_train_step(lf, rf, p, md, gt, dtype=torch.float64)
    lfg, rfg = lf.to(dev).requires_grad_(True), rf.to(dev).requires_grad_(True)
    results, loss_dict = model(dict(leftFeature=lfg, rightFeature=rfg, leftDisp=gt.to(dev)))
    assert results == {} and sorted(loss_dict) == sorted(l64)
    for k, v in l64.items():
        assert abs(loss_dict[k].item() - v.item()) <= 1e-4 * max(1.0, abs(v.item())), k
    sum(loss_dict.values()).backward()
    named = dict(model.named_parameters())
    tight = 0
    zero = 1e-6 * max(v.abs().max().item() for v in g64.values())   # the convolution biases in front of BatchNorm: exactly zero
    for k, ref in g64.items():
        got = lfg.grad if k == "ref_fms" else rfg.grad if k == "tgt_fms" else named[k].grad
        assert got is not None, k
        scale = ref.abs().max().item()
        err = (got.cpu().double() - ref).abs().max().item()
        assert err <= 3e-2 * scale + zero, "grad of %s: error %.3e of range %.3e" % (k, err, scale)
        tight += err <= 4 * (g32[k].double() - ref).abs().max().item() + 2e-5 * scale + zero
    assert len(g64) == 80 + 7 + 3 and tight >= 0.6 * len(g64)   # PSMNet's 80 + 7 convolution biases + 3 up-sampling kernels


@pytest.mark.parametrize("Ci,Co,k,dil,shape", [
    (192, 64, 3, 1, (1, 20, 64)),
    (32, 32, 3, 1, (2, 9, 72)),          # partial strip in x, two batch items
    (48, 10, 3, 1, (1, 33, 132)),        # channel counts that are not multiples of 32, several row segments
    (16, 16, 3, 1, (1, 5, 8)),
    (128, 128, 3, 2, (1, 17, 68)),       # dilation 2 (backbone layer4): six ring slots
    (20, 40, 3, 2, (2, 6, 12)),
    (32, 32, 3, 4, (1, 19, 72)),         # dilations of the refinement blocks: rows walked in interleaved classes
    (32, 32, 3, 8, (2, 21, 64)),
    (32, 16, 3, 8, (1, 6, 20)),          # fewer rows than the dilation
    (64, 128, 1, 1, (1, 12, 64)),        # 1x1 convolutions (down-sampling shortcuts, SPP branches, lastconv)
    (320, 32, 1, 1, (2, 7, 20)),
    (128, 32, 1, 1, (2, 2, 3)),          # SPP branch sizes: width not a multiple of 4 (padded by the wrapper)
    (24, 24, 3, 1, (1, 6, 10)),
])
def test_conv2d_wgrad_and_dgrad(dev, Ci, Co, k, dil, shape):
    """2-D weight / data gradients (AcfNet's confidence heads, the stride-1 layers of the 2-D networks) against autograd of
    F.conv2d."""
    import torch.nn.functional as F
    ops = _ops()
    B, H, W = shape
    x, dc, w = _rand((B, Ci, H, W), 51), _rand((B, Co, H, W), 52), _rand((Co, Ci, k, k), 53, 0.05)
    res = {}
    for dt in (torch.float32, torch.float64):
        xr, wr = x.to(dt).requires_grad_(True), w.to(dt).requires_grad_(True)
        res[dt] = torch.autograd.grad(F.conv2d(xr, wr, None, padding=dil * (k // 2), dilation=dil), (xr, wr), dc.to(dt))
    dw = ops.conv2d_wgrad(x.to(dev), dc.to(dev), k, dil).cpu()
    _close(dw, res[torch.float64][1], res[torch.float32][1], "dW (2-D)")
    dx = ops.conv2d_dgrad(dc.to(dev), w.to(dev), dil).cpu()
    _close(dx, res[torch.float64][0], res[torch.float32][0], "dx (2-D)")


def test_channel_dot(dev):
    ops = _ops()
    a, g = _rand((3, 20, 7, 13), 61), _rand((3, 1, 7, 13), 62)
    ref = (a.double() * g.double()).sum(dim=(0, 2, 3))
    got = ops.channel_dot(a.to(dev), g.to(dev)).cpu()
    assert (got.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


def test_acfnet_adaptive_training_step(dev):
    """AcfNet with its confidence network in training mode (configs/AcfNet/scene_flow_adaptive.py): the heads' NLL loss, the
    focal loss with the per-pixel variance they supply (whose gradient flows back into them), smooth-L1; every gradient
    against the oracle.  Tolerances as in test_psmnet_training_step."""
    import os
    from densematchingbenchmark_amd.config import Config
    from densematchingbenchmark_amd.modeling import build_model
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", "AcfNet", "scene_flow_adaptive.py"))
    md = 32
    cfg.model.max_disp = md
    cfg.model.cost_processor.cost_computation.max_disp = md // 4
    cfg.model.cost_processor.cost_aggregator.max_disp = md
    cfg.model.disp_predictor.max_disp = md
    cfg.model.losses.l1_loss.max_disp = md
    cfg.model.losses.focal_loss.max_disp = md
    cfg.model.cmn.in_planes = md
    cfg.model.cmn.losses.nll_loss.max_disp = md
    p = O.with_prefix(O.random_params_psm(seed=4, classif_gain=4.0, acf=True), "cost_processor.aggregator.")
    g = torch.Generator().manual_seed(77)
    Cm = md // 3
    for i in range(3):
        pre = "cmn.conf_heads.%d.conf_net." % i
        p[pre + "0.0.weight"] = (torch.rand((Cm, md, 3, 3), generator=g) * 2 - 1) / (md * 9) ** 0.5
        p[pre + "0.1.weight"] = 0.5 + torch.rand(Cm, generator=g)
        p[pre + "0.1.bias"] = (torch.rand(Cm, generator=g) - 0.5) * 0.2
        p[pre + "0.1.running_mean"] = torch.zeros(Cm)
        p[pre + "0.1.running_var"] = torch.ones(Cm)
        p[pre + "1.weight"] = (torch.rand((1, Cm, 1, 1), generator=g) * 2 - 1) / Cm ** 0.5
    model = build_model(cfg, backbone=None)
    missing, unexpected = model.load_state_dict(p, strict=False)
    assert not unexpected
    model = model.to(dev).train()
    lf, rf = _rand((2, 32, 8, 24), 41), _rand((2, 32, 8, 24), 42)
    gt = torch.rand((2, 1, 32, 96), generator=torch.Generator().manual_seed(43)) * 40.0 - 4.0
    l32, g32, run32 = O.acfnet_train_step(lf, rf, p, md, gt, adaptive=True)
    l64, g64, _ = O.acfnet_train_step(lf, rf, p, md, gt, adaptive=True, dtype=torch.float64)
    lfg, rfg = lf.to(dev).requires_grad_(True), rf.to(dev).requires_grad_(True)
    results, loss_dict = model(dict(leftFeature=lfg, rightFeature=rfg, leftDisp=gt.to(dev)))
    assert results == {} and sorted(loss_dict) == sorted(l64)
    for k, v in l64.items():
        assert abs(loss_dict[k].item() - v.item()) <= 1e-4 * max(1.0, abs(v.item())), k
    sum(loss_dict.values()).backward()
    named = dict(model.named_parameters())
    tight = 0
    zero = 1e-6 * max(v.abs().max().item() for v in g64.values())
    for k, ref in g64.items():
        got = lfg.grad if k == "ref_fms" else rfg.grad if k == "tgt_fms" else named[k].grad
        assert got is not None, k
        scale = ref.abs().max().item()
        err = (got.cpu().double() - ref).abs().max().item()
        assert err <= 3e-2 * scale + zero, "grad of %s: error %.3e of range %.3e" % (k, err, scale)
        tight += err <= 4 * (g32[k].double() - ref).abs().max().item() + 2e-5 * scale + zero
    assert len(g64) == 90 + 3 * 4 and tight >= 0.6 * len(g64)   # + (conv, gamma, beta, 1x1 conv) per confidence head
    buffers = dict(model.named_buffers())
    for k, v in run32.items():
        assert (buffers[k].cpu() - v).abs().max().item() <= 1e-5 * max(1.0, v.abs().max().item()), k


def test_avgpool_and_bilinear_backward(dev):
    import torch.nn.functional as F
    ops = _ops()
    x = _rand((2, 5, 20, 37), 71)
    for k in (8, 16):
        xr = x.clone().double().requires_grad_(True)
        y = F.avg_pool2d(xr, (k, k), stride=(k, k))
        dy = _rand(tuple(y.shape), 72)
        ref, = torch.autograd.grad(y, xr, dy.double())
        got = ops.avgpool2d_bwd(dy.to(dev), (20, 37), k).cpu()
        assert (got.double() - ref).abs().max().item() <= 1e-7
    lo = _rand((2, 6, 3, 4), 73)
    lr = lo.clone().double().requires_grad_(True)
    up = F.interpolate(lr, (17, 30), mode="bilinear", align_corners=True)
    dy = _rand(tuple(up.shape), 74)
    ref, = torch.autograd.grad(up, lr, dy.double())
    got = ops.bilinear_ac_bwd(dy.to(dev), (3, 4)).cpu()
    assert (got.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


def test_psmnet_backbone_training(dev):
    """The PSMNet backbone in train() mode (both views, BatchNorm statistics per view, SPP branches): features and the gradient
    of every parameter against the oracle's autograd.  Tolerances as in test_psmnet_training_step."""
    from densematchingbenchmark_amd import synthetic
    from densematchingbenchmark_amd.modeling.stereo.backbones import PSMNetBackbone
    bb = PSMNetBackbone(3, True)
    synthetic.init_params_(bb, seed=8, classif_gain=1.0)
    p = {"backbone." + k: v.clone() for k, v in bb.state_dict().items()}
    bb = bb.to(dev).train()
    li, ri = _rand((2, 3, 256, 320), 81), _rand((2, 3, 256, 320), 82)   # (2 images per view: BatchNorm needs > 1 value after the 64x64 pooling)
    dl, dr = _rand((2, 32, 64, 80), 83), _rand((2, 32, 64, 80), 84)
    (fl32, fr32), g32, run32 = O.psmnet_backbone_train_step(li, ri, p, dl, dr)
    (fl64, fr64), g64, _ = O.psmnet_backbone_train_step(li, ri, p, dl, dr, dtype=torch.float64)
    fl, fr = bb(li.to(dev), ri.to(dev))
    assert (fl.detach().cpu().double() - fl64).abs().max().item() <= 2e-5 and (fr.detach().cpu().double() - fr64).abs().max().item() <= 2e-5
    ((fl * dl.to(dev)).sum() + (fr * dr.to(dev)).sum()).backward()
    named = dict(bb.named_parameters())
    tight = 0
    zero = 1e-6 * max(v.abs().max().item() for v in g64.values())
    for k, ref in g64.items():
        got = named[k[len("backbone."):]].grad
        assert got is not None, k
        scale = ref.abs().max().item()
        err = (got.cpu().double() - ref).abs().max().item()
        assert err <= 3e-2 * scale + zero, "grad of %s: error %.3e of range %.3e" % (k, err, scale)
        tight += err <= 4 * (g32[k].double() - ref).abs().max().item() + 2e-5 * scale + zero
    assert tight >= 0.6 * len(g64), "only %d of %d gradients within the tight tolerance" % (tight, len(g64))
    buffers = dict(bb.named_buffers())
    for k, v in run32.items():
        assert (buffers[k[len("backbone."):]].cpu() - v).abs().max().item() <= 1e-5 * max(1.0, v.abs().max().item()), k


def test_eval_mode_gradients(dev):
    """model.eval() with an input that requires a gradient (fine-tuning with frozen statistics, saliency): BatchNorm uses its
    running buffers, which stay untouched, and the gradients are those of the eval-mode graph."""
    from densematchingbenchmark_amd.modeling.stereo.cost_processors.aggregators import PSMAggregator
    from densematchingbenchmark_amd.modeling.stereo.cost_processors.utils.cat_fms import cat_fms
    from densematchingbenchmark_amd.modeling.stereo.disp_predictors import PREDICTORS
    from densematchingbenchmark_amd.modeling.stereo.losses import DispSmoothL1Loss
    md = 32
    p = O.random_params_psm(seed=9, classif_gain=4.0)
    agg = PSMAggregator(max_disp=md, in_planes=64, batch_norm=True)
    agg.load_state_dict(p, strict=False)
    agg = agg.to(dev).eval()
    before = {k: v.clone() for k, v in agg.named_buffers()}
    pred = PREDICTORS['FASTER'](max_disp=md).to(dev)
    lf, rf = _rand((1, 32, 8, 24), 91), _rand((1, 32, 8, 24), 92)
    gt = torch.rand((1, 1, 32, 96), generator=torch.Generator().manual_seed(93)) * 30.0 + 1.0
    pp = O.with_prefix(p, "cost_processor.aggregator.")
    _, g32, _ = O.psmnet_train_step(lf, rf, pp, md, gt, training=False)
    _, g64, _ = O.psmnet_train_step(lf, rf, pp, md, gt, training=False, dtype=torch.float64)
    lfg, rfg = lf.to(dev).requires_grad_(True), rf.to(dev).requires_grad_(True)
    costs = agg(cat_fms(lfg, rfg, md // 4, 0, 1))
    losses = DispSmoothL1Loss(max_disp=md, weights=(1.0, 0.7, 0.5))([pred(c) for c in costs], gt.to(dev))
    sum(losses.values()).backward()
    for k, v in agg.named_buffers():
        assert torch.equal(v, before[k]), k      # eval mode: no running-statistics update
    named = dict(agg.named_parameters())

    def check():
        tight = 0
        for k, ref in g64.items():
            got = lfg.grad if k == "ref_fms" else rfg.grad if k == "tgt_fms" else named[k[len("cost_processor.aggregator."):]].grad
            assert got is not None, k
            scale = ref.abs().max().item()
            err = (got.cpu().double() - ref).abs().max().item()
            assert err <= 3e-2 * scale, "grad of %s: error %.3e of range %.3e" % (k, err, scale)
            tight += err <= 4 * (g32[k].double() - ref).abs().max().item() + 2e-5 * scale
        assert tight >= 0.6 * len(g64)

    check()
    # fine-tuning with frozen statistics: the model in train(), its BatchNorm layers in eval() -- the same graph again, from
    # plain inputs this time (a training-mode unit builds the graph for its parameters)
    agg.zero_grad(set_to_none=True)
    agg.train()
    for m in agg.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()
    lfg, rfg = lf.to(dev).requires_grad_(True), rf.to(dev).requires_grad_(True)
    costs = agg(cat_fms(lfg, rfg, md // 4, 0, 1))
    sum(DispSmoothL1Loss(max_disp=md, weights=(1.0, 0.7, 0.5))([pred(c) for c in costs], gt.to(dev)).values()).backward()
    for k, v in agg.named_buffers():
        assert torch.equal(v, before[k]), k
    check()


def test_bilinear_scale_backward(dev):
    import torch.nn.functional as F
    ops = _ops()
    lo = _rand((2, 1, 6, 9), 95)
    lr = lo.clone().double().requires_grad_(True)
    up = F.interpolate(lr, (48, 72), mode="bilinear", align_corners=False) * 8.0
    dy = _rand(tuple(up.shape), 96)
    ref, = torch.autograd.grad(up, lr, dy.double())
    got = ops.bilinear_scale_bwd(dy.to(dev), (6, 9), 8.0).cpu()
    assert (got.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


def test_stereonet_end_to_end_training(dev):
    """The whole StereoNet-8x model in train() mode, images -> losses: backbone (5x5 stride-2 heads through the space-to-depth
    form, BasicBlocks), difference volume, aggregator, soft-argmin, edge-aware refinement (dilations 1, 2, 4, 8; half-pixel
    up-sampling; 1-channel residual head), smooth-L1 on the refined and the coarse map; every gradient against the oracle.
    Tolerances as in test_psmnet_training_step."""
    import os
    from densematchingbenchmark_amd import synthetic
    from densematchingbenchmark_amd.config import Config
    from densematchingbenchmark_amd.modeling import build_model
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", "StereoNet", "scene_flow_8x_2stage.py"))
    cfg.model.disp_refinement = dict(type="StereoNet", in_planes=4, num=1)
    cfg.model.backbone = dict(type="StereoNet", in_planes=3)
    cfg.model.losses = dict(l1_loss=dict(max_disp=192, weights=(1.0, 0.5), weight=1.0))
    model = build_model(cfg, backbone="hip")
    synthetic.init_params_(model, seed=12, classif_gain=4.0)
    p = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev).train()
    li, ri = _rand((2, 3, 64, 96), 97), _rand((2, 3, 64, 96), 98)
    gt = torch.rand((2, 1, 64, 96), generator=torch.Generator().manual_seed(99)) * 30.0 + 0.5
    l32, g32, run32 = O.stereonet_e2e_train_step(li, ri, p, 192, gt)
    l64, g64, _ = O.stereonet_e2e_train_step(li, ri, p, 192, gt, dtype=torch.float64)
    results, loss_dict = model(dict(leftImage=li.to(dev), rightImage=ri.to(dev), leftDisp=gt.to(dev)))
    assert results == {} and sorted(loss_dict) == ["l1_loss_lvl0", "l1_loss_lvl1"]
    for i in range(2):
        assert abs(loss_dict["l1_loss_lvl%d" % i].item() - l64[i].item()) <= 1e-4 * max(1.0, abs(l64[i].item()))
    sum(loss_dict.values()).backward()
    named = dict(model.named_parameters())
    tight, checked = 0, 0
    zero = 1e-6 * max(v.abs().max().item() for v in g64.values() if v is not None)
    for k, ref in g64.items():
        if ref is None:
            continue
        got = named[k].grad
        assert got is not None, k
        scale = ref.abs().max().item()
        err = (got.cpu().double() - ref).abs().max().item()
        assert err <= 3e-2 * scale + zero, "grad of %s: error %.3e of range %.3e" % (k, err, scale)
        tight += err <= 4 * (g32[k].double() - ref).abs().max().item() + 2e-5 * scale + zero
        checked += 1
    assert checked >= 100 and tight >= 0.6 * checked


def test_gcnet_end_to_end_training(dev):
    """The whole GC-Net model, images -> loss -> gradients: 5x5 stride-2 first layer (space-to-depth form), BasicBlocks,
    concatenation volume at 1/2 resolution, the 3-D encoder / decoder with 64 / 128-channel units (chunked stride-2 data
    gradients), ReLU-before-skip transposed units, the 1-channel transposed head.

    At a size the CPU oracle can differentiate in FP64, GC-Net's deepest level holds 1 x 2 x 3 voxels per channel: batch
    statistics over 12 values make the training-mode gradients ill-conditioned (the FP32 oracle itself sits up to 1e-1 of the
    range from its FP64 evaluation).  So the per-gradient bound is checked on the SAME graph with running statistics
    (eval mode, input requiring a gradient), and training mode is checked for its loss and for the direction of every
    gradient (cosine similarity with the FP64 oracle)."""
    import os
    from densematchingbenchmark_amd import synthetic
    from densematchingbenchmark_amd.config import Config
    from densematchingbenchmark_amd.modeling import build_model
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", "GCNet", "scene_flow.py"))
    md = 32
    cfg.model.max_disp = md
    cfg.model.cost_processor.cost_computation.max_disp = md // 2
    cfg.model.cost_processor.cost_aggregator.max_disp = md
    cfg.model.disp_predictor.max_disp = md
    cfg.model.losses = dict(l1_loss=dict(max_disp=md, weights=(1.0,), weight=1.0))
    model = build_model(cfg, backbone="hip")
    synthetic.init_params_(model, seed=14, classif_gain=4.0)
    p = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev)
    li, ri = _rand((2, 3, 64, 96), 101), _rand((2, 3, 64, 96), 102)
    gt = torch.rand((2, 1, 64, 96), generator=torch.Generator().manual_seed(103)) * 28.0 + 0.5
    named = dict(model.named_parameters())

    # (a) running statistics: the per-gradient bounds of test_psmnet_training_step
    _, g32, _ = O.gcnet_e2e_train_step(li, ri, p, md, gt, training=False)
    l64, g64, _ = O.gcnet_e2e_train_step(li, ri, p, md, gt, training=False, dtype=torch.float64)
    model.eval()
    from densematchingbenchmark_amd.modeling.stereo.losses import DispSmoothL1Loss
    # (an eval-mode module builds a graph only for inputs that carry a gradient: both views must)
    lig, rig = li.to(dev).requires_grad_(True), ri.to(dev).requires_grad_(True)
    fl, fr = model.backbone(lig, rig)
    disp = model.disp_predictor(model.cost_processor(fl, fr)[0])
    loss = DispSmoothL1Loss(max_disp=md, weights=(1.0,))(disp, gt.to(dev))["l1_loss_lvl0"]
    assert abs(loss.item() - l64.item()) <= 1e-4 * max(1.0, abs(l64.item()))
    loss.backward()
    tight, checked = 0, 0
    # (the head's bias shifts every cost of a pixel alike: its gradient is exactly zero, rounding noise in any evaluation)
    zero = 1e-5 * max(v.abs().max().item() for v in g64.values() if v is not None)
    for k, ref in g64.items():
        if ref is None:
            continue
        got = named[k].grad
        assert got is not None, k
        scale = ref.abs().max().item()
        err = (got.cpu().double() - ref).abs().max().item()
        assert err <= 3e-2 * scale + zero, "grad of %s: error %.3e of range %.3e" % (k, err, scale)
        tight += err <= 4 * (g32[k].double() - ref).abs().max().item() + 2e-5 * scale + zero
        checked += 1
    assert checked >= 100 and tight >= 0.6 * checked

    # (b) batch statistics: loss and gradient directions
    model.zero_grad(set_to_none=True)
    model.train()
    l64, g64, _ = O.gcnet_e2e_train_step(li, ri, p, md, gt, dtype=torch.float64)
    results, loss_dict = model(dict(leftImage=li.to(dev), rightImage=ri.to(dev), leftDisp=gt.to(dev)))
    assert results == {} and list(loss_dict) == ["l1_loss_lvl0"]
    assert abs(loss_dict["l1_loss_lvl0"].item() - l64.item()) <= 1e-4 * max(1.0, abs(l64.item()))
    sum(loss_dict.values()).backward()
    for k, ref in g64.items():
        if ref is None or ref.abs().max().item() < 1e-12:
            continue
        got = named[k].grad.cpu().double().reshape(-1)
        assert torch.isfinite(got).all(), k
        cos = torch.dot(got, ref.reshape(-1)) / (got.norm() * ref.norm() + 1e-300)
        assert cos.item() >= 0.98, "grad of %s: cosine %.4f" % (k, cos.item())


def test_train_mode_without_autograd_uses_batch_statistics(dev):
    """train() under torch.no_grad() (a forward-only pass in training mode): batch statistics and running-buffer updates are
    training-mode semantics, not autograd's -- the result must be the training-mode forward, not the folded inference one."""
    from densematchingbenchmark_amd.modeling.stereo.cost_processors.aggregators import AGGREGATORS
    from tests._util import golden
    g = golden("aggregators.npz")
    p = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("snp_")}
    agg = AGGREGATORS["StereoNet"](max_disp=12, in_planes=32, batch_norm=True, num=4)
    agg.load_state_dict(p, strict=False)
    agg = agg.to(dev).train()
    raw = _rand((2, 32, 12, 10, 28), 111)
    q = {k: v.clone() for k, v in p.items()}
    with O.bn_training():
        want = O.stereonet_aggregator(raw, q, "", num=4)[0]
    with torch.no_grad():
        got = agg(raw.to(dev))[0]
    assert not got.requires_grad
    assert (got.cpu() - want).abs().max().item() <= 2e-5
    buffers = dict(agg.named_buffers())
    for k, v in q.items():
        if "running_" in k:
            assert (buffers[k].cpu() - v).abs().max().item() <= 1e-5 * max(1.0, v.abs().max().item()), k


def test_eval_after_training_mode_forward_sees_the_new_running_statistics(dev):
    """eval -> train() under no_grad -> eval: the training-mode kernel rewrites the running buffers through raw pointers
    (their tensor version does not move), so the eval-path fold cache is keyed on num_batches_tracked as well.  The second
    eval forward must equal the oracle's eval forward on the UPDATED buffers, not the first one."""
    from densematchingbenchmark_amd.modeling.stereo.cost_processors.aggregators import AGGREGATORS
    from tests._util import golden
    g = golden("aggregators.npz")
    p = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("snp_")}
    agg = AGGREGATORS["StereoNet"](max_disp=12, in_planes=32, batch_norm=True, num=4)
    agg.load_state_dict(p, strict=False)
    agg = agg.to(dev).eval()
    raw = _rand((2, 32, 12, 10, 28), 111)
    with torch.no_grad():
        first = agg(raw.to(dev))[0].cpu()
        agg.train()
        agg(raw.to(dev) * 1.7 + 0.3)          # moves every running buffer
        agg.eval()
        second = agg(raw.to(dev))[0].cpu()
    q = {k: v.detach().cpu().clone() for k, v in agg.state_dict().items()}
    want = O.stereonet_aggregator(raw, q, "", num=4)[0]
    assert (first - second).abs().max().item() > 1e-3          # the statistics did change the result
    assert (second - want).abs().max().item() <= 2e-5


def test_eval_mode_module_with_trainable_weights_gets_gradients(dev):
    """``model.train(); unit.eval()`` (frozen BatchNorm statistics): the unit's weights still receive gradients, its
    BatchNorm normalises with the running buffers and leaves them untouched -- as with the reference's plain nn modules."""
    from densematchingbenchmark_amd.modeling.stereo.layers.basic_layers import conv3d_bn_relu
    unit = conv3d_bn_relu(True, 8, 32, 3, 1, 1, bias=False).to(dev)
    with torch.no_grad():
        unit[1].running_mean.uniform_(-0.2, 0.2)
        unit[1].running_var.uniform_(0.5, 1.5)
    unit.eval()
    x = _rand((1, 8, 4, 6, 16), 5).to(dev)
    rm = unit[1].running_mean.clone()
    y = unit(x)                                                    # grad enabled, plain input, trainable weights
    assert y.requires_grad
    y.square().mean().backward()
    assert unit[0].weight.grad is not None and torch.isfinite(unit[0].weight.grad).all() and unit[0].weight.grad.abs().max() > 0
    assert torch.equal(unit[1].running_mean, rm)                   # eval-mode BatchNorm: buffers untouched
    ref = torch.nn.Sequential(torch.nn.Conv3d(8, 32, 3, 1, 1, bias=False), torch.nn.BatchNorm3d(32), torch.nn.ReLU()).eval()
    ref.load_state_dict({k: v.detach().cpu() for k, v in unit.state_dict().items()})
    yr = ref(x.cpu())
    yr.square().mean().backward()
    assert (y.detach().cpu() - yr.detach()).abs().max().item() <= 2e-5
    assert (unit[0].weight.grad.cpu() - ref[0].weight.grad).abs().max().item() <= 1e-5 * max(1.0, ref[0].weight.grad.abs().max().item()) + 1e-6
    with torch.no_grad():                                          # and under no_grad the fused inference kernel
        assert not unit(x).requires_grad


@pytest.mark.parametrize("shape", [(2, 8, 5, 6, 12), (4, 32, 6, 16, 32), (3, 64, 3, 5, 9), (2, 5, 33, 20), (2, 64, 40, 64)])
@pytest.mark.parametrize("relu,with_res", [(0, False), (1, False), (1, True), (2, True)])
def test_bn_train_fwd_is_the_two_entry_points_in_two_launches(dev, shape, relu, with_res):
    """dmb_bn_train_fwd_f32 (ABI 8: block sums + ONE kernel that finishes the statistics, updates the running buffers and the batch
    counter and normalises) against dmb_bn_train_stats_f32 + dmb_bn_act_f32: the same arithmetic, so every output bit for bit."""
    ops = _ops()
    C = shape[1]
    c = (_rand(shape, 21) * 1.5 + 0.3).to(dev)
    gamma, beta = (_rand((C,), 22) * 0.5 + 1.0).to(dev), (_rand((C,), 23) * 0.2).to(dev)
    res = _rand(shape, 24).to(dev) if with_res else None
    rm0, rv0 = _rand((C,), 26) * 0.1, _rand((C,), 27).abs() + 0.5
    mode = {0: False, 1: True, 2: "pre"}[relu]
    rm1, rv1 = rm0.to(dev), rv0.to(dev)
    mean, invstd, scale, shift = ops.bn_train_stats(c, gamma, beta, rm1, rv1, momentum=0.1, eps=1e-5)
    y1 = ops.bn_act(c, scale, shift, res, relu=mode)
    rm2, rv2 = rm0.to(dev), rv0.to(dev)
    nbt = torch.tensor(5, dtype=torch.int64, device=dev)
    y2, mean2, invstd2, scale2, shift2 = ops.bn_train_fwd(c, gamma, beta, rm2, rv2, nbt, 0.1, 1e-5, res, mode)
    for a, b, what in ((y1, y2, "y"), (mean, mean2, "mean"), (invstd, invstd2, "invstd"), (scale, scale2, "scale"), (shift, shift2, "shift"),
                       (rm1, rm2, "running_mean"), (rv1, rv2, "running_var")):
        assert torch.equal(a, b), what
    assert int(nbt) == 6
    # without affine parameters, running buffers or a counter
    y3, mean3, _, scale3, _ = ops.bn_train_fwd(c, None, None, None, None, None, 0.1, 1e-5, res, mode)
    m3, i3, s3, h3 = ops.bn_train_stats(c, None, None, None, None, momentum=0.1, eps=1e-5)
    assert torch.equal(y3, ops.bn_act(c, s3, h3, res, relu=mode)) and torch.equal(mean3, m3) and torch.equal(scale3, s3)


@pytest.mark.parametrize("shape", [(2, 8, 5, 6, 12), (3, 32, 4, 7, 9), (2, 5, 33, 20)])
@pytest.mark.parametrize("relu", [0, 1, 2])
@pytest.mark.parametrize("training", [True, False])
def test_bn_act_bwd_accumulating_skip_gradient(dev, shape, relu, training):
    """``dres_acc`` of dmb_bn_act_bwd_f32 (ABI 8): the skip branch's gradient comes back as (its own share) + (what the skip
    operand already holds) -- exactly the FP32 sum autograd's own addition launch would have written; dc / dgamma / dbeta unchanged."""
    ops = _ops()
    C = shape[1]
    c, dy, res, acc = (_rand(shape, 31) + 0.2).to(dev), _rand(shape, 32).to(dev), _rand(shape, 33).to(dev), _rand(shape, 34).to(dev)
    gamma, beta = (_rand((C,), 35) * 0.5 + 1.0).to(dev), (_rand((C,), 36) * 0.2).to(dev)
    mode = {0: False, 1: True, 2: "pre"}[relu]
    mean, invstd, scale, shift = ops.bn_train_stats(c, gamma, beta, None, None, momentum=0.1, eps=1e-5)
    y = ops.bn_act(c, scale, shift, res, relu=mode)
    dc0, dg0, db0, dres0 = ops.bn_act_bwd(dy, c, y, scale, shift, mean, invstd, relu=mode, training=training, want_dres=True)
    dc1, dg1, db1, dres1 = ops.bn_act_bwd(dy, c, y, scale, shift, mean, invstd, relu=mode, training=training, dres_acc=acc)
    assert torch.equal(dc0, dc1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert torch.equal(dres1, dres0 + acc)
    if relu != 1:
        assert torch.equal(dres0, dy)


def test_data_gradients_take_the_collected_gradient_as_skip_operand(dev):
    """conv3d_k3_dgrad / deconv3d_k3s2_dgrad / conv2d_dgrad with ``residual``: the convolution's result + the operand, added in the
    epilogue (scale 1, shift 0: fma(acc, 1, 0) + r), i.e. exactly what a separate addition of the two tensors gives."""
    ops = _ops()
    w = _rand((32, 32, 3, 3, 3), 41, 0.1).to(dev)
    dc = _rand((2, 32, 4, 6, 16), 42).to(dev)
    r = _rand((2, 32, 4, 6, 16), 43).to(dev)
    assert torch.equal(ops.conv3d_k3_dgrad(dc, w, 1, residual=r), ops.conv3d_k3_dgrad(dc, w, 1) + r)
    w2 = _rand((64, 32, 3, 3, 3), 44, 0.1).to(dev)          # a stride-2 unit 32 -> 64: its adjoint is the transposed kernel
    dc2 = _rand((2, 64, 2, 3, 8), 45).to(dev)
    assert torch.equal(ops.conv3d_k3_dgrad(dc2, w2, 2, (4, 6, 16), residual=r), ops.conv3d_k3_dgrad(dc2, w2, 2, (4, 6, 16)) + r)
    r_odd = _rand((2, 32, 3, 5, 15), 46).to(dev)            # odd extents: the cropped adjoint, the addition outside the kernel
    assert torch.equal(ops.conv3d_k3_dgrad(dc2, w2, 2, (3, 5, 15), residual=r_odd), ops.conv3d_k3_dgrad(dc2, w2, 2, (3, 5, 15)) + r_odd)
    wt = _rand((64, 32, 3, 3, 3), 47, 0.1).to(dev)          # a transposed unit 64 -> 32: its adjoint is the stride-2 kernel
    dyt = _rand((2, 32, 4, 6, 16), 48).to(dev)
    rt = _rand((2, 64, 2, 3, 8), 49).to(dev)
    assert torch.equal(ops.deconv3d_k3s2_dgrad(dyt, wt, residual=rt), ops.deconv3d_k3s2_dgrad(dyt, wt) + rt)
    w2d = _rand((32, 160, 3, 3), 50, 0.1).to(dev)           # 2-D, more than one 128-channel chunk of the result
    d2, r2 = _rand((2, 32, 9, 20), 51).to(dev), _rand((2, 160, 9, 20), 52).to(dev)
    assert torch.equal(ops.conv2d_dgrad(d2, w2d, 1, residual=r2), ops.conv2d_dgrad(d2, w2d, 1) + r2)


def test_gradient_carry_equals_autograd_sums(dev):
    """train_fn's gradient carry (the sums over a tensor's consumers formed inside the consumers' backward kernels) against
    torch.autograd's own additions: same losses bit for bit (the forward does not change), every gradient equal up to the order of
    a handful of FP32 additions (<= 2e-6 of its range), and far fewer addition launches."""
    import os
    from densematchingbenchmark_amd.config import Config
    from densematchingbenchmark_amd.modeling import build_model
    from densematchingbenchmark_amd.modeling.stereo.layers import train_fn
    from densematchingbenchmark_amd import synthetic
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", "PSMNet", "scene_flow.py"))
    md = 32
    cfg.model.max_disp = md
    cfg.model.cost_processor.cost_computation.max_disp = md // 4
    cfg.model.cost_processor.cost_aggregator.max_disp = md
    cfg.model.disp_predictor.max_disp = md
    cfg.model.losses.l1_loss.max_disp = md
    model = build_model(cfg, backbone=None).to(dev)
    synthetic.init_params_(model, seed=3)
    model.train()
    lf, rf = _rand((2, 32, 8, 24), 61).to(dev), _rand((2, 32, 8, 24), 62).to(dev)
    gt = (torch.rand((2, 1, 32, 96), generator=torch.Generator().manual_seed(63)) * 30.0 + 1.0).to(dev)
    state = {k: v.clone() for k, v in model.state_dict().items()}

    def run(flag):
        train_fn.set_gradient_carry(flag)
        try:
            model.load_state_dict(state)                       # the running buffers start from the same values both times
            model.zero_grad(set_to_none=True)
            a, b = lf.clone().requires_grad_(True), rf.clone().requires_grad_(True)
            _, losses = model(dict(leftFeature=a, rightFeature=b, leftDisp=gt))
            with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
                sum(losses.values()).backward()
                torch.cuda.synchronize()
            adds = sum(e.count for e in prof.key_averages() if "CUDAFunctor_add<float>" in e.key)
            grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
            grads["ref_fms"], grads["tgt_fms"] = a.grad.clone(), b.grad.clone()
            return [float(v) for v in losses.values()], grads, adds
        finally:
            train_fn.set_gradient_carry(True)

    l_on, g_on, adds_on = run(True)
    l_off, g_off, adds_off = run(False)
    assert l_on == l_off
    assert set(g_on) == set(g_off) and len(g_on) == 80
    for k in g_on:
        scale = g_off[k].abs().max().item()
        assert (g_on[k] - g_off[k]).abs().max().item() <= 2e-6 * scale + 1e-30, k
    # autograd's own additions: 18 tensors with several consumers without the carry (cost0 x 3, out1 / out2, pre / post of the
    # hourglasses, ...); with it only the two the chain cannot reach remain (cost1 / cost2 feed a head's skip AND the regression)
    assert adds_off >= adds_on + 10, (adds_on, adds_off)


def test_first_unit_without_the_volume_in_training(dev):
    """The training path's first unit on a concatenation volume (train_fn.CatConvUnitFn: 2-D form forward, volume only inside the
    backward) against the materialised volume + ConvUnitFn (``ops.set_cat_fusion(False)``): losses within 1e-5 relative, every
    gradient within 1e-4 of its range (the two forwards differ by FP32 summation order; everything downstream sees that)."""
    import os
    from densematchingbenchmark_amd.config import Config
    from densematchingbenchmark_amd.modeling import build_model
    from densematchingbenchmark_amd import ops, synthetic
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", "PSMNet", "scene_flow.py"))
    md = 32
    cfg.model.max_disp = md
    cfg.model.cost_processor.cost_computation.max_disp = md // 4
    cfg.model.cost_processor.cost_aggregator.max_disp = md
    cfg.model.disp_predictor.max_disp = md
    cfg.model.losses.l1_loss.max_disp = md
    model = build_model(cfg, backbone=None).to(dev)
    synthetic.init_params_(model, seed=5)
    model.train()
    lf, rf = _rand((2, 32, 8, 24), 71).to(dev), _rand((2, 32, 8, 24), 72).to(dev)
    gt = (torch.rand((2, 1, 32, 96), generator=torch.Generator().manual_seed(73)) * 30.0 + 1.0).to(dev)
    state = {k: v.clone() for k, v in model.state_dict().items()}

    def run(flag):
        ops.set_cat_fusion(flag)
        try:
            model.load_state_dict(state)
            model.zero_grad(set_to_none=True)
            a, b = lf.clone().requires_grad_(True), rf.clone().requires_grad_(True)
            with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
                _, losses = model(dict(leftFeature=a, rightFeature=b, leftDisp=gt))
                torch.cuda.synchronize()
            volumes = sum(e.count for e in prof.key_averages() if "volume_kernel" in e.key)
            sum(losses.values()).backward()
            grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
            grads["ref_fms"], grads["tgt_fms"] = a.grad.clone(), b.grad.clone()
            return [float(v) for v in losses.values()], grads, volumes
        finally:
            ops.set_cat_fusion(True)

    l_on, g_on, vol_on = run(True)
    l_off, g_off, vol_off = run(False)
    assert vol_on == 0 and vol_off == 1          # the forward pass of the fused form never writes the volume
    for a, b in zip(l_on, l_off):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b))
    assert set(g_on) == set(g_off) and len(g_on) == 80
    worst = 0.0
    for k in g_on:
        scale = g_off[k].abs().max().item()
        worst = max(worst, (g_on[k] - g_off[k]).abs().max().item() / (scale + 1e-30))
    # a ReLU within rounding of zero may flip between the two forwards (see test_psmnet_training_step): 3e-2 of the range at worst,
    # and without such a flip (this seed) 1e-4
    assert worst <= 3e-2, worst
    print("first unit 2-D form vs materialised volume: worst gradient difference %.2e of its range" % worst)


def test_pack_group_repacks_every_unit_in_one_launch(dev):
    """train_fn._PackGroup (dmb_conv3d_pack_weights_multi_f32, ABI 8): the forward and data-gradient packs of all units made by one
    launch per optimizer step -- bit for bit the packs of the per-unit entry points, so losses and gradients are identical to
    ``set_pack_group(False)``; a second step after an in-place weight update re-packs (and sees the new weights)."""
    import os
    from densematchingbenchmark_amd.config import Config
    from densematchingbenchmark_amd.modeling import build_model
    from densematchingbenchmark_amd.modeling.stereo.layers import train_fn
    from densematchingbenchmark_amd import ops, synthetic
    # the table's three modes against the single-tensor entry points
    w = _rand((64, 32, 3, 3, 3), 81).to(dev)
    for transposed, stride, refs in ((False, 1, (ops.pack_conv3d_weights, ops.pack_conv3d_dgrad_weights)),
                                     (False, 2, (ops.pack_conv3d_weights, ops.pack_deconv3d_weights)),
                                     (True, 2, (ops.pack_deconv3d_weights, ops.pack_conv3d_weights))):
        jobs = ops.unit_pack_jobs(w, transposed, stride)
        bufs = [torch.full((ops.packed_floats(co, ci),), 7.0, device=dev) for co, ci, _ in jobs]
        table = ops.make_pack_table([(w, b, co, ci, m) for (co, ci, m), b in zip(jobs, bufs)], w.device)
        ops.run_pack_table(table, len(jobs))
        for b, ref in zip(bufs, refs):
            assert torch.equal(b, ref(w)), (transposed, stride)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", "PSMNet", "scene_flow.py"))
    md = 32
    cfg.model.max_disp = md
    cfg.model.cost_processor.cost_computation.max_disp = md // 4
    cfg.model.cost_processor.cost_aggregator.max_disp = md
    cfg.model.disp_predictor.max_disp = md
    cfg.model.losses.l1_loss.max_disp = md
    model = build_model(cfg, backbone=None).to(dev)
    synthetic.init_params_(model, seed=9)
    model.train()
    lf, rf = _rand((2, 32, 8, 24), 82).to(dev), _rand((2, 32, 8, 24), 83).to(dev)
    gt = (torch.rand((2, 1, 32, 96), generator=torch.Generator().manual_seed(84)) * 30.0 + 1.0).to(dev)
    state = {k: v.clone() for k, v in model.state_dict().items()}

    def two_steps(flag):
        train_fn.set_pack_group(flag)
        try:
            model.load_state_dict(state)
            out = []
            packs = 0
            for step in range(3):
                model.zero_grad(set_to_none=True)
                with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
                    _, losses = model(dict(leftFeature=lf, rightFeature=rf, leftDisp=gt))
                    sum(losses.values()).backward()
                    torch.cuda.synchronize()
                if step > 0:   # (the first step registers the units one by one: a re-pack per newcomer)
                    packs += sum(e.count for e in prof.key_averages() if "pack_weights" in e.key and ("multi" in e.key) == flag)
                out.append(([float(v.detach()) for v in losses.values()], {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
                with torch.no_grad():
                    for p in model.parameters():
                        if p.grad is not None:
                            p.sub_(1e-3 * p.grad)      # an in-place update, as an optimizer does
            return out, packs
        finally:
            train_fn.set_pack_group(True)

    on, packs_on = two_steps(True)
    off, packs_off = two_steps(False)
    assert packs_on == 2 and packs_off >= 100, (packs_on, packs_off)     # one launch per step against two per unit and step
    for (la, ga), (lb, gb) in zip(on, off):
        assert la == lb
        assert all(torch.equal(ga[k], gb[k]) for k in ga)
    assert on[0][0] != on[1][0] != on[2][0]                                           # the second step did see the updated weights


def test_fused_optimizer_updates_are_seen(dev):
    """torch's fused optimizers update parameters WITHOUT moving their ``_version`` (checked here), which every packed / folded
    parameter cache of this package keys on.  The training path's pack group therefore re-packs once per forward pass whatever the
    versions say, and a train() -> eval() switch advances ops' parameter epoch: (a) two Adam(fused=True) steps give the same losses
    with the pack group as with per-call packing, bit for bit; (b) the eval-mode forward after them equals a fresh model loaded
    with the trained state."""
    import os
    from densematchingbenchmark_amd.config import Config
    from densematchingbenchmark_amd.modeling import build_model
    from densematchingbenchmark_amd.modeling.stereo.layers import train_fn
    from densematchingbenchmark_amd import synthetic
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", "PSMNet", "scene_flow.py"))
    md = 32
    cfg.model.max_disp = md
    cfg.model.cost_processor.cost_computation.max_disp = md // 4
    cfg.model.cost_processor.cost_aggregator.max_disp = md
    cfg.model.disp_predictor.max_disp = md
    cfg.model.losses.l1_loss.max_disp = md
    lf, rf = _rand((2, 32, 8, 24), 91).to(dev), _rand((2, 32, 8, 24), 92).to(dev)
    gt = (torch.rand((2, 1, 32, 96), generator=torch.Generator().manual_seed(93)) * 30.0 + 1.0).to(dev)

    def run(flag):
        train_fn.set_pack_group(flag)
        try:
            model = build_model(cfg, backbone=None).to(dev)
            synthetic.init_params_(model, seed=11)
            model.eval()
            with torch.no_grad():
                model(dict(leftFeature=lf, rightFeature=rf))          # fills the eval path's caches with the INITIAL weights
            model.train()
            params = [p for p in model.parameters() if p.requires_grad]
            opt = torch.optim.Adam(params, lr=1e-2, fused=True)
            losses = []
            for _ in range(3):
                opt.zero_grad(set_to_none=True)
                _, ld = model(dict(leftFeature=lf, rightFeature=rf, leftDisp=gt))
                sum(ld.values()).backward()
                v = params[0]._version
                opt.step()
                assert params[0]._version == v                         # the premise: a fused step leaves the version alone
                losses.append([float(x.detach()) for x in ld.values()])
            model.eval()
            with torch.no_grad():
                out, _ = model(dict(leftFeature=lf, rightFeature=rf))
            return losses, out["disps"][0].clone(), {k: v.clone() for k, v in model.state_dict().items()}
        finally:
            train_fn.set_pack_group(True)

    l_on, d_on, state = run(True)
    l_off, d_off, _ = run(False)
    assert l_on == l_off and l_on[0] != l_on[1] != l_on[2]
    assert torch.equal(d_on, d_off)
    fresh = build_model(cfg, backbone=None).to(dev)
    fresh.load_state_dict(state)
    fresh.eval()
    with torch.no_grad():
        want, _ = fresh(dict(leftFeature=lf, rightFeature=rf))
    assert torch.equal(d_on, want["disps"][0])


@pytest.mark.parametrize("kind", ["cat", "dif"])
@pytest.mark.parametrize("B,C,Co,D,H,W", [(2, 8, 32, 8, 6, 24), (1, 32, 32, 12, 5, 36), (2, 5, 16, 4, 7, 16), (1, 4, 8, 16, 3, 20)])
def test_first_unit_weight_gradient_without_the_volume(dev, kind, B, C, Co, D, H, W):
    """ops.cat_first_wgrad (dmb_cat_first_wgrad_maps_f32 + two 2-D weight gradients) against the 3-D weight gradient on the
    materialised volume and against torch's CPU autograd of conv3d(cat_fms(L, R)) in FP64: the z fold and the validity masks
    (volume zero where x < z, zero padding around it, D larger than / close to W) are exact index work, the sums FP32."""
    ops = _ops()
    L, R = _rand((B, C, H, W), 101), _rand((B, C, H, W), 102)
    dc = _rand((B, Co, D, H, W), 103)
    idx = list(range(D))
    vol = O.cat_fms(L, R, D, 0, 1) if kind == "cat" else O.dif_fms(L, R, D, 0, 1)
    w = torch.zeros((Co, vol.shape[1], 3, 3, 3), dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv3d(vol.double(), w, padding=1).backward(dc.double())
    ref64 = w.grad
    w32 = torch.zeros((Co, vol.shape[1], 3, 3, 3), requires_grad=True)
    torch.nn.functional.conv3d(vol, w32, padding=1).backward(dc)
    got = ops.cat_first_wgrad(L.to(dev), R.to(dev), dc.to(dev), kind)
    assert got.shape == ref64.shape
    _close(got.cpu(), ref64, w32.grad, "first-unit weight gradient (%s)" % kind)
    volg = ops.cat_fms(L.to(dev), R.to(dev), idx) if kind == "cat" else ops.dif_fms(L.to(dev), R.to(dev), idx)
    _close(ops.conv3d_k3_wgrad(volg, dc.to(dev)).cpu(), ref64, w32.grad, "3-D weight gradient on the volume (%s)" % kind)


@pytest.mark.parametrize("transposed", [False, True])
def test_s2_weight_gradient_tiles_at_crop_sized_volumes(dev, transposed):
    """dmb_conv3d_k3s2_wgrad_f32 picks its tile per launch (round 6: 2 x 12 or 4 x 8 voxels of the small tensor).  A volume with
    64-column rows and several rounds of items takes the 4 x 8 tile (checked by kernel name), a launch of a single round the 2 x 12
    tile; both against torch's FP64 autograd evaluated on the GPU (checker only; the CPU oracle would take minutes at this size),
    within the tolerance an FP32 sum over 260 k voxels is held to elsewhere in this file."""
    ops = _ops()
    import torch.nn.functional as F
    for (Ds, Hs, Ws), want in (((8, 32, 64), "Wg2Cfg<4, 8>"), ((4, 8, 24), "Wg2Cfg<2, 12>")):
        big = _rand((2, 32, 2 * Ds, 2 * Hs, 2 * Ws), 111).to(dev)
        small = _rand((2, 64, Ds, Hs, Ws), 112).to(dev)
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
            got = ops.deconv3d_k3s2_wgrad(small, big) if transposed else ops.conv3d_k3s2_wgrad(big, small)
            torch.cuda.synchronize()
        names = [e.key for e in prof.key_averages() if "conv3d_wgrad_s2_kernel" in e.key]
        assert len(names) == 1 and want in names[0], names
        if transposed:     # ConvTranspose3d(64 -> 32): x = small, dy = big, weight [64, 32, 27]
            w = torch.zeros((64, 32, 3, 3, 3), dtype=torch.float64, device=dev, requires_grad=True)
            F.conv_transpose3d(small.double(), w, stride=2, padding=1, output_padding=1).backward(big.double())
        else:              # Conv3d(32 -> 64, stride 2): x = big, dc = small, weight [64, 32, 27]
            w = torch.zeros((64, 32, 3, 3, 3), dtype=torch.float64, device=dev, requires_grad=True)
            F.conv3d(big.double(), w, stride=2, padding=1).backward(small.double())
        ref = w.grad
        scale = ref.abs().max().item()
        assert (got.double() - ref).abs().max().item() <= 2e-5 * scale, (Ws, transposed)


@pytest.mark.parametrize("shape", [(2, 32, 8, 12, 48), (1, 32, 5, 9, 32), (2, 16, 6, 7, 24), (4, 32, 12, 16, 128), (1, 64, 4, 6, 60)])
def test_batch_statistics_from_the_convolution_epilogue(dev, shape):
    """dmb_conv3d_k3_bnstats_f32 + dmb_bn_train_act_f32 (ABI 8) against the convolution followed by dmb_bn_train_fwd_f32: the raw
    output bit for bit; the partial sums against FP64 sums of that output (partial tiles in every direction are masked); mean /
    invstd / scale / shift / running buffers within 1e-6 relative (the two forms sum in a different order and the fused one has no
    pivot) and the normalised output within 2e-6 of its range."""
    ops = _ops()
    B, Ci, D, H, W = shape
    x = _rand(shape, 121).to(dev)
    w = (_rand((32, Ci, 3, 3, 3), 122) * 0.1).to(dev)
    wp = ops.pack_conv3d_weights(w)
    gamma, beta = (_rand((32,), 123) * 0.5 + 1.0).to(dev), (_rand((32,), 124) * 0.2).to(dev)
    res = _rand((B, 32, D, H, W), 125).to(dev)
    fused = ops.conv3d_k3_bnstats(x, wp, 32)
    assert fused is not None
    raw, parts = fused
    ops.set_split_k(False)       # (the reference launch on the single-chain kernels too: small launches otherwise take the split-K form)
    try:
        ref_raw = ops.conv3d_k3(x, wp, 32)
    finally:
        ops.set_split_k(True)
    assert torch.equal(raw, ref_raw)
    tot = parts.sum(dim=1)
    r64 = ref_raw.double()
    # (the four voxels of a 16-byte word are added, and squared, in FP32 before they enter the FP64 sums: 1e-7 of a word's value each)
    n = r64[:, 0].numel()
    assert torch.allclose(tot[:, 0], r64.sum(dim=(0, 2, 3, 4)), rtol=1e-7, atol=3e-7 * n ** 0.5 * float(r64.abs().max()))
    assert torch.allclose(tot[:, 1], (r64 * r64).sum(dim=(0, 2, 3, 4)), rtol=1e-6, atol=1e-6)
    rm0, rv0 = _rand((32,), 126) * 0.1, _rand((32,), 127).abs() + 0.5
    rm1, rv1, rm2, rv2 = rm0.to(dev), rv0.to(dev), rm0.to(dev), rv0.to(dev)
    n1, n2 = torch.tensor(3, dtype=torch.int64, device=dev), torch.tensor(3, dtype=torch.int64, device=dev)
    y1, m1, i1, s1, h1 = ops.bn_train_fwd(ref_raw, gamma, beta, rm1, rv1, n1, 0.1, 1e-5, res, True)
    y2, m2, i2, s2, h2 = ops.bn_train_act(raw, parts, gamma, beta, rm2, rv2, n2, 0.1, 1e-5, res, True)
    assert int(n1) == int(n2) == 4
    for a, b, what in ((m1, m2, "mean"), (i1, i2, "invstd"), (s1, s2, "scale"), (h1, h2, "shift"), (rm1, rm2, "running_mean"), (rv1, rv2, "running_var")):
        assert (a - b).abs().max().item() <= 1e-6 * max(1.0, b.abs().max().item()) + 2e-7, what
    assert (y1 - y2).abs().max().item() <= 2e-6 * max(1.0, y1.abs().max().item())
    # an output with a large mean: the pivot-free sums still give the variance (FP64)
    big = raw + 300.0
    parts_big = torch.stack([big.double().sum(dim=(0, 2, 3, 4)), (big.double() ** 2).sum(dim=(0, 2, 3, 4))], -1).unsqueeze(1).contiguous()
    _, mb, ib, _, _ = ops.bn_train_act(big, parts_big, None, None, None, None, None, 0.1, 1e-5, None, False)
    _, mr, ir, _, _ = ops.bn_train_fwd(big, None, None, None, None, None, 0.1, 1e-5, None, False)
    assert (mb - mr).abs().max().item() <= 1e-4 and (ib - ir).abs().max().item() <= 1e-4 * ir.abs().max().item()


def test_epilogue_statistics_in_a_training_step(dev):
    """A PSMNet training iteration with the batch statistics of the 32-channel stride-1 units taken from the convolution epilogues
    against the same iteration with a pass per unit (``set_epilogue_stats(False)``): losses within 1e-5 relative, gradients within
    1e-4 of their range (the statistics differ in the last bits, everything downstream with them), no bn_stats launch for those units."""
    import os
    from densematchingbenchmark_amd.config import Config
    from densematchingbenchmark_amd.modeling import build_model
    from densematchingbenchmark_amd.modeling.stereo.layers import train_fn
    from densematchingbenchmark_amd import synthetic
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", "PSMNet", "scene_flow.py"))
    md = 32
    cfg.model.max_disp = md
    cfg.model.cost_processor.cost_computation.max_disp = md // 4
    cfg.model.cost_processor.cost_aggregator.max_disp = md
    cfg.model.disp_predictor.max_disp = md
    cfg.model.losses.l1_loss.max_disp = md
    model = build_model(cfg, backbone=None).to(dev)
    synthetic.init_params_(model, seed=13)
    model.train()
    lf, rf = _rand((2, 32, 8, 24), 131).to(dev), _rand((2, 32, 8, 24), 132).to(dev)
    gt = (torch.rand((2, 1, 32, 96), generator=torch.Generator().manual_seed(133)) * 30.0 + 1.0).to(dev)
    state = {k: v.clone() for k, v in model.state_dict().items()}

    def run(flag):
        train_fn.set_epilogue_stats(flag)
        try:
            model.load_state_dict(state)
            model.zero_grad(set_to_none=True)
            with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
                _, losses = model(dict(leftFeature=lf, rightFeature=rf, leftDisp=gt))
                torch.cuda.synchronize()
            nstats = sum(e.count for e in prof.key_averages() if "bn_stats_kernel" in e.key)
            sum(losses.values()).backward()
            return ([float(v.detach()) for v in losses.values()], {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None},
                    nstats, {k: v.clone() for k, v in model.state_dict().items() if "running" in k})
        finally:
            train_fn.set_epilogue_stats(False)      # (the library's default: the fused form does not pay, docs/design/12-6)

    l_on, g_on, n_on, r_on = run(True)
    l_off, g_off, n_off, r_off = run(False)
    assert n_off == 25 and n_on == 25 - 6        # dres0[1], dres1[0], dres1[1], classif1-3[0]: the 32-channel stride-1 units on a tensor input
    for a, b in zip(l_on, l_off):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b))
    for k in g_on:
        assert (g_on[k] - g_off[k]).abs().max().item() <= 3e-2 * g_off[k].abs().max().item() + 1e-30, k
    for k in r_on:
        assert (r_on[k] - r_off[k]).abs().max().item() <= 1e-5 * max(1.0, r_off[k].abs().max().item()), k


def test_flat_gradients_gather_on_the_device(dev):
    """dist_utils.FlatGradients in its default mode on device tensors: a PSMNet training iteration leaves one fresh gradient tensor
    per parameter (no accumulation launches), gather_() packs them with a multi-tensor copy into the flat buffer and every grad
    becomes its view; a second iteration without zero_() accumulates into the views."""
    import os
    from densematchingbenchmark_amd.config import Config
    from densematchingbenchmark_amd.dist_utils import FlatGradients
    from densematchingbenchmark_amd.modeling import build_model
    from densematchingbenchmark_amd import synthetic
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", "PSMNet", "scene_flow.py"))
    md = 32
    cfg.model.max_disp = md
    cfg.model.cost_processor.cost_computation.max_disp = md // 4
    cfg.model.cost_processor.cost_aggregator.max_disp = md
    cfg.model.disp_predictor.max_disp = md
    cfg.model.losses.l1_loss.max_disp = md
    model = build_model(cfg, backbone=None).to(dev)
    synthetic.init_params_(model, seed=17)
    model.train()
    for m in model.modules():                      # frozen statistics: two iterations on the same batch give the same gradients
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()
    flat = FlatGradients(model)
    lf, rf = _rand((1, 32, 8, 24), 141).to(dev), _rand((1, 32, 8, 24), 142).to(dev)
    gt = (torch.rand((1, 1, 32, 96), generator=torch.Generator().manual_seed(143)) * 30.0 + 1.0).to(dev)
    flat.zero_()
    assert all(p.grad is None for p in flat.params)
    _, losses = model(dict(leftFeature=lf, rightFeature=rf, leftDisp=gt))
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        sum(losses.values()).backward()
        torch.cuda.synchronize()
    assert sum(e.count for e in prof.key_averages() if "CUDAFunctor_add<float>" in e.key) <= 6   # no `grad += new` per parameter
    assert not flat.attached() and all(p.grad is not None for p in flat.params)
    ref = [p.grad.clone() for p in flat.params]
    flat.gather_()
    assert flat.attached() and all(torch.equal(p.grad, r) for p, r in zip(flat.params, ref))
    _, losses = model(dict(leftFeature=lf, rightFeature=rf, leftDisp=gt))
    sum(losses.values()).backward()
    assert flat.attached()
    for p, r in zip(flat.params, ref):
        assert (p.grad - 2 * r).abs().max().item() <= 1e-5 * max(1e-6, r.abs().max().item()) + 1e-12


def test_fused_optimizer_updates_are_seen_by_the_backbone(dev):
    """The same for the 2-D units (phase-lived weight packs, two views per step): three Adam(fused=True) steps of the whole PSMNet
    model on small images give the same losses as per-call packing, bit for bit, and every step sees the previous update."""
    import os
    from densematchingbenchmark_amd.config import Config
    from densematchingbenchmark_amd.modeling import build_model
    from densematchingbenchmark_amd.modeling.stereo.layers import train_fn
    from densematchingbenchmark_amd import synthetic
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", "PSMNet", "scene_flow.py"))
    md = 32
    cfg.model.max_disp = md
    cfg.model.cost_processor.cost_computation.max_disp = md // 4
    cfg.model.cost_processor.cost_aggregator.max_disp = md
    cfg.model.disp_predictor.max_disp = md
    cfg.model.losses.l1_loss.max_disp = md
    g = torch.Generator().manual_seed(151)
    # (256 x 512: the smallest image the backbone's 64 x 64 pooling branch admits, backbones/PSMNet.py:43)
    li, ri = torch.randn((1, 3, 256, 512), generator=g).to(dev), torch.randn((1, 3, 256, 512), generator=g).to(dev)
    gt = (torch.rand((1, 1, 256, 512), generator=g) * 30.0 + 1.0).to(dev)

    def run(flag):
        train_fn.set_pack_group(flag)
        try:
            model = build_model(cfg, backbone="hip").to(dev)
            synthetic.init_params_(model, seed=19)
            model.train()
            opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3, fused=True)
            losses = []
            for _ in range(3):
                opt.zero_grad(set_to_none=True)
                _, ld = model(dict(leftImage=li, rightImage=ri, leftDisp=gt))
                sum(ld.values()).backward()
                opt.step()
                losses.append([float(x.detach()) for x in ld.values()])
            return losses
        finally:
            train_fn.set_pack_group(True)

    on, off = run(True), run(False)
    assert on == off and on[0] != on[1] != on[2]
