"""Generate tests/golden/fullsize_*.npz: the REAL reference (imported read-only from /root/reference, recipe of
oracle/gen_golden.py) run on CPU at the BASELINE sizes -- PSMNet 544x960 / max_disp 192 (all four pairs of a
bench batch), AcfNet 544x960 (one pair, with its confidence network), StereoNet-8x 384x1248 (one pair) -- on the
seeded parameters and inputs the GPU tests regenerate.  Stored: sub-sampled disparity / confidence maps and cost rows
(data, not code).  Also pins the oracle at full size (tests/test_oracle_golden.py checks its FP32 path against the same
files at the sizes the CPU suite can afford).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_fullsize.py          (about five minutes on 8 cores)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import gen_golden as G  # noqa: E402  (import recipe, config loader)

OUT = G.OUT
SUB = (slice(None), slice(None), slice(3, None, 8), slice(5, None, 8))      # maps [B, 1, H, W] -> every 8th pixel
CROWS = (slice(None), slice(7, None, 48), slice(11, None, 136), slice(None))  # costs [B, D, H, W] -> 4 planes x 4 rows


class _M(torch.nn.Module):
    pass


def main():
    G.import_reference()
    torch.set_num_threads(int(os.environ.get("DMB_THREADS", "8")))
    from dmb.modeling.stereo.cost_processors import build_cost_processor
    from dmb.modeling.stereo.disp_predictors import build_disp_predictor
    from dmb.modeling.stereo.cmn.cmn import Cmn, ConfHead
    from densematchingbenchmark_amd import synthetic   # seeded parameters / inputs only (shared with the GPU tests)

    with torch.no_grad():
        # ---- PSMNet, BASELINE configs[1]: 544x960, max_disp 192, the four pairs of rank 0's bench batch -------------
        cfg = G.load_cfg("configs/PSMNet/scene_flow.py")
        m = _M()
        m.cost_processor = build_cost_processor(cfg)
        m.disp_predictor = build_disp_predictor(cfg)
        m.eval()
        synthetic.init_params_(m, seed=0, classif_gain=10.0)
        out = {}
        for i in range(4):
            lf, rf = synthetic.feature_pair(i, 32, 136, 240)
            costs = m.cost_processor(lf, rf)
            disps = [m.disp_predictor(c) for c in costs]
            for lvl, (d, c) in enumerate(zip(disps, costs)):
                out["pair%d_disp%d" % (i, 3 - lvl)] = G.npy(d[SUB])
                if i == 0:
                    out["pair0_cost%d_rows" % (3 - lvl)] = G.npy(c[CROWS])
            print("psmnet pair", i, "disp3 range %.2f..%.2f" % (disps[0].min().item(), disps[0].max().item()), flush=True)
            del costs, disps
        np.savez_compressed(os.path.join(OUT, "fullsize_psmnet.npz"), **out)

        # ---- AcfNet, BASELINE configs[3]: 544x960, one pair, adaptive config with the confidence network --------------
        cfg = G.load_cfg("configs/AcfNet/scene_flow_adaptive.py")
        m = _M()
        m.cost_processor = build_cost_processor(cfg)
        m.disp_predictor = build_disp_predictor(cfg)
        m.cmn = Cmn.__new__(Cmn)
        torch.nn.Module.__init__(m.cmn)
        m.cmn.conf_heads = torch.nn.ModuleList([ConfHead(cfg.model.cmn.in_planes, True) for _ in range(3)])
        m.cmn.alpha, m.cmn.beta = cfg.model.cmn.alpha, cfg.model.cmn.beta
        m.eval()
        synthetic.init_params_(m, seed=5, classif_gain=10.0)
        lf, rf = synthetic.feature_pair(0, 32, 136, 240)
        costs = m.cost_processor(lf, rf)
        disps = [m.disp_predictor(c) for c in costs]
        confs, cost_vars, _ = Cmn.get_confidence(m.cmn, costs)
        out = {}
        for lvl in range(3):
            out["disp%d" % (3 - lvl)] = G.npy(disps[lvl][SUB])
            out["conf%d" % (3 - lvl)] = G.npy(confs[lvl][SUB])
            out["var%d" % (3 - lvl)] = G.npy(cost_vars[lvl][SUB])
            out["cost%d_rows" % (3 - lvl)] = G.npy(costs[lvl][CROWS])
        print("acfnet disp3 range %.2f..%.2f conf3 range %.3f..%.3f" % (disps[0].min().item(), disps[0].max().item(),
                                                                     confs[0].min().item(), confs[0].max().item()), flush=True)
        np.savez_compressed(os.path.join(OUT, "fullsize_acfnet.npz"), **out)
        del costs, disps, confs, cost_vars

        # ---- StereoNet-8x, BASELINE configs[4]: 384x1248 (375x1242 padded), one pair, cost path at 1/8 ----------------
        cfg = G.load_cfg("configs/StereoNet/scene_flow_8x_2stage.py")
        m = _M()
        m.cost_processor = build_cost_processor(cfg)
        m.disp_predictor = build_disp_predictor(cfg)
        m.eval()
        synthetic.init_params_(m, seed=6, classif_gain=10.0)
        lf, rf = synthetic.feature_pair(0, 32, 48, 156)
        costs = m.cost_processor(lf, rf)
        d = m.disp_predictor(costs[0])
        np.savez_compressed(os.path.join(OUT, "fullsize_stereonet.npz"), disp=G.npy(d), cost=G.npy(costs[0][:, :, 1::2, :]))
        print("stereonet disp range %.2f..%.2f" % (d.min().item(), d.max().item()), flush=True)
    for f in sorted(os.listdir(OUT)):
        if f.startswith("fullsize"):
            print("%-28s %8.1f KB" % (f, os.path.getsize(os.path.join(OUT, f)) / 1024))


if __name__ == "__main__":
    main()
