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


def _close(got, ref64, ref32, what):
    """|got - FP64| must stay within 4x the FP32 oracle's own error (plus a floor of 2e-6 of the value range)."""
    scale = ref64.abs().max().item()
    err = (got.double() - ref64).abs().max().item()
    own = (ref32.double() - ref64).abs().max().item()
    assert err <= 4 * own + 2e-6 * scale, "%s: error %.3e vs oracle FP32 error %.3e (range %.3e)" % (what, err, own, scale)


@pytest.mark.parametrize("Ci,Co,shape", [
    (32, 32, (1, 5, 8, 24)),
    (32, 32, (2, 9, 7, 28)),        # partial tiles in y and x, two z segments
    (64, 32, (1, 4, 9, 48)),
    (32, 64, (1, 6, 6, 24)),
    (8, 40, (1, 3, 5, 12)),         # channel counts that are not multiples of 32
    (32, 32, (1, 4, 6, 30)),        # W % 4 != 0 -> dword staging
    (32, 32, (1, 20, 12, 48)),
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
])
def test_conv3d_dgrad(dev, Ci, Co, stride, shape):
    ops = _ops()
    B, D, H, W = shape
    x = _rand((B, Ci, D, H, W), 1)
    w = _rand((Co, Ci, 3, 3, 3), 3, 0.05)
    dc = _rand((B, Co, (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1), 2)
    dx32, _ = O.conv3d_backward(x, w, dc, stride)
    dx64, _ = O.conv3d_backward(x, w, dc, stride, dtype=torch.float64)
    got = ops.conv3d_k3_dgrad(dc.to(dev), w.to(dev), stride).cpu()
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
