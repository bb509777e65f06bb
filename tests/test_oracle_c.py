"""Cross-check of the two halves of the oracle on small shapes: the plain-C loop nests (oracle/dmb_oracle_c.c)
against the PyTorch-CPU restatement (oracle/dmb_oracle.py, itself pinned to the real reference by golden vectors).
Independent index arithmetic, so an error in either shows up here.  CPU only."""
import ctypes
import math
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dmb_oracle as O
from tests._util import rand

ODIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


@pytest.fixture(scope="module")
def oc():
    subprocess.check_call(["make", "-s", "-C", ODIR])
    return ctypes.CDLL(os.path.join(ODIR, "_build", "libdmb_oracle_c.so"))


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _ints(v):
    return (ctypes.c_int * len(v))(*v)


def test_volumes(oc):
    L, R = rand((2, 6, 5, 17), 1), rand((2, 6, 5, 17), 2)
    for md, sd, dil in ((7, -3, 1), (6, 0, 2), (20, 0, 1)):
        idx = O.disp_index_list(md, sd, dil)
        out = torch.empty(2, 12, len(idx), 5, 17)
        oc.oc_cat_fms(_p(L), _p(R), _p(out), 2, 6, 5, 17, len(idx), _ints(idx))
        assert torch.equal(out, O.cat_fms(L, R, md, sd, dil))
        out = torch.empty(2, 6, len(idx), 5, 17)
        oc.oc_dif_fms(_p(L), _p(R), _p(out), 2, 6, 5, 17, len(idx), _ints(idx))
        assert torch.equal(out, O.dif_fms(L, R, md, sd, dil))
        out = torch.empty(2, 3, len(idx), 5, 17)
        oc.oc_gwc_fms(_p(L), _p(R), _p(out), 2, 6, 3, 5, 17, len(idx), _ints(idx))
        assert (out - O.gwc_fms(L, R, md, sd, dil, 3)).abs().max().item() <= 1e-6


@pytest.mark.parametrize("stride", [1, 2])
def test_conv3d(oc, stride):
    x, w = rand((2, 4, 5, 6, 7), 3), rand((3, 4, 3, 3, 3), 4, 0.2)
    sc, sh = torch.rand(3) + 0.5, torch.rand(3) - 0.5
    ref = F.conv3d(x, w, None, stride=stride, padding=1) * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1)
    res = rand(ref.shape, 5)
    y = torch.empty_like(ref)
    oc.oc_conv3d_k3(_p(x), _p(w), _p(sc), _p(sh), _p(res), _p(y), 2, 4, 3, 5, 6, 7, stride, 1)
    assert (y - F.relu(ref + res)).abs().max().item() <= 2e-6


def test_deconvs(oc):
    x, w = rand((2, 4, 3, 4, 5), 6), rand((4, 3, 3, 3, 3), 7, 0.2)
    ref = F.conv_transpose3d(x, w, None, stride=2, padding=1, output_padding=1)
    y = torch.empty_like(ref)
    oc.oc_deconv3d_k3s2(_p(x), _p(w), None, None, None, _p(y), 2, 4, 3, 3, 4, 5, 0)
    assert y.shape == (2, 3, 6, 8, 10) and (y - ref).abs().max().item() <= 2e-6
    x1, w1 = rand((2, 3, 4, 5), 8), rand((8, 8, 8), 9, 0.1)
    ref = F.conv_transpose3d(x1.unsqueeze(1), w1.view(1, 1, 8, 8, 8), None, stride=4, padding=2).squeeze(1)
    y = torch.empty_like(ref)
    oc.oc_deconv3d_k8s4_c1(_p(x1), _p(w1), _p(y), 2, 3, 4, 5)
    assert (y - ref).abs().max().item() <= 2e-6


def test_trilinear(oc):
    x = rand((2, 4, 6, 10), 10)
    for outs in ((16, 24, 40), (11, 17, 30)):
        ref = F.interpolate(x.unsqueeze(1), list(outs), mode="trilinear", align_corners=True).squeeze(1)
        y = torch.empty_like(ref)
        oc.oc_trilinear_ac(_p(x), _p(y), 2, 4, 6, 10, *outs)
        assert (y - ref).abs().max().item() <= 1e-6


def test_predictors_and_errors(oc):
    cost = rand((2, 24, 5, 7), 11, 6.0)
    vals = O.disp_sample_values(24)
    y = torch.empty(2, 1, 5, 7)
    oc.oc_soft_argmin(_p(cost), _p(y), 2, 24, 5, 7, ctypes.c_float(1.0), 1, _p(vals))
    assert (y - O.soft_argmin_f64(cost, 24).float()).abs().max().item() <= 2e-6
    idx = torch.empty(2, 1, 5, 7, dtype=torch.int64)
    oc.oc_local_soft_argmin(_p(cost), _p(y), _p(idx), 2, 24, 5, 7, 2, 1, 0, 1, ctypes.c_float(1.0))
    ref, ridx = O.local_soft_argmin(cost, 24, 2)
    assert torch.equal(idx, ridx) and (y - ref).abs().max().item() <= 1e-5
    g = torch.Generator().manual_seed(12)
    gt = torch.rand((1, 1, 20, 32), generator=g) * 220 - 10
    est = gt + torch.randn((1, 1, 20, 32), generator=g) * 3
    out = (ctypes.c_double * 5)()
    oc.oc_calc_error(_p(est), _p(gt), 20, 32, 17, 30, ctypes.c_float(0), ctypes.c_float(192), out)
    e = O.calc_error(O.remove_padding(est, (17, 30)), O.remove_padding(gt, (17, 30)), 0, 192)
    assert np.allclose(list(out), [e[k] for k in ("epe", "1px", "2px", "3px", "5px")], rtol=1e-6)
