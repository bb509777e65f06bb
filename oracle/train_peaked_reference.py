"""Weights whose cost distributions are PEAKED the way a trained network's are, made by briefly TRAINING the
real reference (imported read-only from /root/reference, recipe of oracle/gen_golden.py) on CPU: its own Concatenation cost
processor + PSMNet aggregator + FasterSoftArgmin (configs/PSMNet/scene_flow.py:19-52), torch autograd, Adam, smooth-L1 on the three
outputs (weights 0.5 / 0.7 / 1.0 as configs/PSMNet/scene_flow.py's losses) -- on synthetic feature pairs with EXACT matches: left
features ~ N(0, 1) per quarter-resolution pixel, right features = the left ones shifted by a banded integer disparity field
(R(y, x') = L(y, x' + d(y)), noise where nothing matches), ground truth 4 d, pixels without a match masked.  A few hundred steps at
64x128 features teach the random-weight network to put one sharp peak at the matching plane; every fixture made from random
weights so far had flat distributions (E|k - disp| = 48 px at D = 192), which maximise the sensitivity of the FP32 soft-argmin chain.

Output: tests/golden/psmnet_trained_weights.npz -- the state_dict rounded to FP16 (inputs of the fixture, not expected outputs:
oracle/gen_golden_fullsize.py `peaked` runs the reference on them at 544x960 and stores what it returns).

    PYTHONDONTWRITEBYTECODE=1 python oracle/train_peaked_reference.py [steps]      (about 3 s per step on 8 cores)
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import gen_golden as G  # noqa: E402


def banded_pair(seed, h, w, planes, bands=4):
    from densematchingbenchmark_amd import synthetic
    return synthetic.banded_match_pair(seed, h, w, planes, bands)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    G.import_reference()
    torch.set_num_threads(int(os.environ.get("DMB_THREADS", "8")))
    from dmb.modeling.stereo.cost_processors import build_cost_processor
    from dmb.modeling.stereo.disp_predictors import build_disp_predictor
    from densematchingbenchmark_amd import synthetic

    cfg = G.load_cfg("configs/PSMNet/scene_flow.py")

    class M(torch.nn.Module):
        pass
    m = M()
    m.cost_processor = build_cost_processor(cfg)
    m.disp_predictor = build_disp_predictor(cfg)
    synthetic.init_params_(m, seed=0, classif_gain=1.0)
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3)
    t0 = time.time()
    for it in range(steps):
        m.train()
        lf, rf, gt = banded_pair(it, 32, 96, 48)
        costs = m.cost_processor(lf, rf)
        disps = [m.disp_predictor(c) for c in costs]
        mask = gt > 0
        loss = sum(wt * F.smooth_l1_loss(d[mask], gt[mask]) for wt, d in zip((1.0, 0.7, 0.5), disps))
        opt.zero_grad()
        loss.backward()
        opt.step()
        if it % 10 == 0 or it == steps - 1:
            with torch.no_grad():
                epe = (disps[0][mask] - gt[mask]).abs().mean().item()
            print("step %4d  loss %.3f  EPE(best level) %.3f px   %.0f s" % (it, loss.item(), epe, time.time() - t0), flush=True)
    m.eval()
    with torch.no_grad():
        lf, rf, gt = banded_pair(99999, 32, 96, 48)
        costs = m.cost_processor(lf, rf)
        d = m.disp_predictor(costs[0])
        pr = torch.softmax(costs[0], 1)
        k = torch.arange(192.).view(1, -1, 1, 1)
        mask = gt > 0
        print("held-out pair, eval mode: EPE %.3f px, cost range %.1f..%.1f, E|k - disp| mean %.2f px, max prob mean %.3f" % (
            (d[mask] - gt[mask]).abs().mean().item(), costs[0].min().item(), costs[0].max().item(),
            (pr * (k - d).abs()).sum(1).mean().item(), pr.max(1)[0].mean().item()), flush=True)
    sd = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    out = {k: (v.astype(np.float16) if v.dtype == np.float32 else v) for k, v in sd.items() if not k.endswith("disp_regression.weight")}
    np.savez_compressed(os.path.join(G.OUT, "psmnet_trained_weights.npz"), **out)
    print("psmnet_trained_weights.npz %.1f KB" % (os.path.getsize(os.path.join(G.OUT, "psmnet_trained_weights.npz")) / 1024))


if __name__ == "__main__":
    main()
