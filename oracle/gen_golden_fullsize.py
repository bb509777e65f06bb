"""Generate tests/golden/fullsize_*.npz: the REAL reference (imported read-only from /root/reference, recipe of
oracle/gen_golden.py) run on CPU at the BASELINE sizes -- PSMNet 544x960 / max_disp 192 (all four pairs of a
bench batch), AcfNet 544x960 (one pair, with its confidence network), StereoNet-8x 384x1248 (one pair) -- on the
seeded parameters and inputs the GPU tests regenerate.  Stored: sub-sampled disparity / confidence maps and cost rows
(data, not code).  Also pins the oracle at full size (tests/test_oracle_golden.py checks its FP32 path against the same
files at the sizes the CPU suite can afford).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_fullsize.py          (about five minutes on 8 cores)
    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_fullsize.py round3   (only the families added in round 3)
    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_fullsize.py kitti    (round 4: the reference's KITTI operating point)
    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_fullsize.py spread   (round 6: the reference against itself at 1 / 3 / 8 threads)
    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_fullsize.py peaked   (round 6: trained weights -> peaked cost distributions)

Round 3 added (parity): PSMNet pair 0 at classifier gain 30 (fullsize_psmnet_gain30.npz); one FULL AcfNet
disparity / confidence map instead of every 64th pixel (fullsize_acfnet_map.npz); and the regression tail -- trilinear x4
up-sampling (PSMNet.py:74-93) + FasterSoftArgmin / SoftArgmin -- on a volume with ground-truth-like peaks near disparity 5
and 185 and costs spanning +-12 (fullsize_regression_ends.npz), which the random-weight networks never produce.
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


def round3():
    """The fixture families added in round 3 (see the module docstring)."""
    import torch.nn.functional as F
    from dmb.modeling.stereo.cost_processors import build_cost_processor
    from dmb.modeling.stereo.disp_predictors import build_disp_predictor
    from dmb.modeling.stereo.disp_predictors.soft_argmin import SoftArgmin
    from dmb.modeling.stereo.disp_predictors.faster_soft_argmin import FasterSoftArgmin
    from dmb.modeling.stereo.cmn.cmn import Cmn, ConfHead
    from densematchingbenchmark_amd import synthetic

    with torch.no_grad():
        # ---- PSMNet pair 0 at classifier gain 30: costs three times as peaked as the gain-10 family ---------------------
        cfg = G.load_cfg("configs/PSMNet/scene_flow.py")
        m = _M()
        m.cost_processor = build_cost_processor(cfg)
        m.disp_predictor = build_disp_predictor(cfg)
        m.eval()
        synthetic.init_params_(m, seed=0, classif_gain=30.0)
        lf, rf = synthetic.feature_pair(0, 32, 136, 240)
        costs = m.cost_processor(lf, rf)
        disps = [m.disp_predictor(c) for c in costs]
        out = {}
        for lvl, (d, c) in enumerate(zip(disps, costs)):
            out["pair0_disp%d" % (3 - lvl)] = G.npy(d[SUB])
            out["pair0_cost%d_rows" % (3 - lvl)] = G.npy(c[CROWS])
        print("psmnet gain 30: disp3 range %.2f..%.2f, cost3 range %.2f..%.2f" % (disps[0].min().item(), disps[0].max().item(),
                                                                                 costs[0].min().item(), costs[0].max().item()), flush=True)
        np.savez_compressed(os.path.join(OUT, "fullsize_psmnet_gain30.npz"), **out)
        del costs, disps

        # ---- AcfNet (as in main(): seed 5, pair 0): the FULL best-level disparity and confidence maps ----------------------
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
        costs = m.cost_processor(lf, rf)
        d3 = m.disp_predictor(costs[0])
        confs, _, _ = Cmn.get_confidence(m.cmn, costs)
        np.savez_compressed(os.path.join(OUT, "fullsize_acfnet_map.npz"), disp3=G.npy(d3), conf3=G.npy(confs[0]).astype(np.float16))
        print("acfnet full maps: disp3 range %.2f..%.2f" % (d3.min().item(), d3.max().item()), flush=True)
        del costs, confs

        # ---- regression tail at the ends of the range: trilinear x4 (PSMNet.py:74-93) + the two soft-argmin modules --------
        q = synthetic.peaked_cost_volume(0, 48, 136, 240)
        full = F.interpolate(q.unsqueeze(1), [192, 544, 960], mode="trilinear", align_corners=True).squeeze(1)
        fast = FasterSoftArgmin(max_disp=192, start_disp=0, dilation=1, alpha=1.0, normalize=True)(full)
        slow = SoftArgmin(max_disp=192, start_disp=0, dilation=1, alpha=1.0, normalize=True)(full)
        print("regression ends: disparity range %.3f..%.3f, cost range %.2f..%.2f, faster-vs-plain %.2e" % (
            fast.min().item(), fast.max().item(), full.min().item(), full.max().item(), (fast - slow).abs().max().item()), flush=True)
        np.savez_compressed(os.path.join(OUT, "fullsize_regression_ends.npz"), faster=G.npy(fast[SUB]), plain=G.npy(slow[SUB]),
                            cost_rows=G.npy(full[CROWS]))


def kitti():
    """Round 4: the reference's PUBLISHED operating point for PSMNet / AcfNet -- KITTI, 384x1248
    (configs/PSMNet/kitti_2015.py:113,122,129; configs/PSMNet/ResultOfPSMNet.md:15-19: 384x1248, 599 ms) -- through the
    reference's OWN KITTI config files: features [1, 32, 96, 312], 48 disparity samples at 1/4 resolution.  Stored: the sampled
    maps and cost rows of every level, plus the FULL best-level disparity map of PSMNet (479 232 pixels).  Also the full
    best-level map of PSMNet pair 0 at 544x960 (a whole map, not every 64th pixel)."""
    from dmb.modeling.stereo.cost_processors import build_cost_processor
    from dmb.modeling.stereo.disp_predictors import build_disp_predictor
    from dmb.modeling.stereo.cmn.cmn import Cmn, ConfHead
    from densematchingbenchmark_amd import synthetic

    crows = (slice(None), slice(7, None, 48), slice(11, None, 24), slice(None))   # costs [1, 192, 384, 1248] -> 4 planes x 16 rows
    with torch.no_grad():
        cfg = G.load_cfg("configs/PSMNet/kitti_2015.py")
        assert list(cfg.data.eval.input_shape) == [384, 1248]
        m = _M()
        m.cost_processor = build_cost_processor(cfg)
        m.disp_predictor = build_disp_predictor(cfg)
        m.eval()
        synthetic.init_params_(m, seed=0, classif_gain=10.0)
        out = {}
        for i in range(2):
            lf, rf = synthetic.feature_pair(i, 32, 96, 312)
            costs = m.cost_processor(lf, rf)
            disps = [m.disp_predictor(c) for c in costs]
            for lvl, (d, c) in enumerate(zip(disps, costs)):
                out["pair%d_disp%d" % (i, 3 - lvl)] = G.npy(d[SUB])
                if i == 0:
                    out["pair0_cost%d_rows" % (3 - lvl)] = G.npy(c[crows])
            if i == 0:
                out["pair0_disp3_full"] = G.npy(disps[0])
            print("psmnet kitti pair", i, "disp3 range %.2f..%.2f" % (disps[0].min().item(), disps[0].max().item()), flush=True)
            del costs, disps
        np.savez_compressed(os.path.join(OUT, "fullsize_psmnet_kitti.npz"), **out)

        cfg = G.load_cfg("configs/AcfNet/kitti_2015_adaptive.py")
        assert list(cfg.data.eval.input_shape) == [384, 1248]
        m = _M()
        m.cost_processor = build_cost_processor(cfg)
        m.disp_predictor = build_disp_predictor(cfg)
        m.cmn = Cmn.__new__(Cmn)
        torch.nn.Module.__init__(m.cmn)
        m.cmn.conf_heads = torch.nn.ModuleList([ConfHead(cfg.model.cmn.in_planes, True) for _ in range(3)])
        m.cmn.alpha, m.cmn.beta = cfg.model.cmn.alpha, cfg.model.cmn.beta
        m.eval()
        synthetic.init_params_(m, seed=5, classif_gain=10.0)
        lf, rf = synthetic.feature_pair(0, 32, 96, 312)
        costs = m.cost_processor(lf, rf)
        disps = [m.disp_predictor(c) for c in costs]
        confs, cost_vars, _ = Cmn.get_confidence(m.cmn, costs)
        out = {}
        for lvl in range(3):
            out["disp%d" % (3 - lvl)] = G.npy(disps[lvl][SUB])
            out["conf%d" % (3 - lvl)] = G.npy(confs[lvl][SUB])
            out["var%d" % (3 - lvl)] = G.npy(cost_vars[lvl][SUB])
            out["cost%d_rows" % (3 - lvl)] = G.npy(costs[lvl][crows])
        print("acfnet kitti disp3 range %.2f..%.2f conf3 range %.3f..%.3f" % (disps[0].min().item(), disps[0].max().item(),
                                                                           confs[0].min().item(), confs[0].max().item()), flush=True)
        np.savez_compressed(os.path.join(OUT, "fullsize_acfnet_kitti.npz"), **out)
        del costs, disps, confs, cost_vars

        # ---- the WHOLE best-level map of PSMNet pair 0 at 544x960 (the sampled fixture holds every 64th pixel) -------------
        cfg = G.load_cfg("configs/PSMNet/scene_flow.py")
        m = _M()
        m.cost_processor = build_cost_processor(cfg)
        m.disp_predictor = build_disp_predictor(cfg)
        m.eval()
        synthetic.init_params_(m, seed=0, classif_gain=10.0)
        lf, rf = synthetic.feature_pair(0, 32, 136, 240)
        costs = m.cost_processor(lf, rf)
        d3 = m.disp_predictor(costs[0])
        np.savez_compressed(os.path.join(OUT, "fullsize_psmnet_map.npz"), disp3=G.npy(d3))
        print("psmnet full map: disp3 range %.2f..%.2f" % (d3.min().item(), d3.max().item()), flush=True)


def spread():
    """Round 6: how far the reference is from ITSELF.  The real reference on pair 0 of BASELINE configs[1]
    (544x960, D = 192) and of the KITTI operating point (384x1248), the same weights and inputs, at 1, 3 and 8 host threads
    (torch.set_num_threads): the convolution back end splits its reductions differently, the costs move by a few 1e-7 and the
    k-ascending FP32 soft-argmin chain (faster_soft_argmin.py:46-71) by up to 1e-4.  Stored: the three sub-sampled map sets of
    every level, the whole-map spread max_ij |ref_i - ref_j| per level (scalars) and the whole best-level maps as differences
    against the 8-thread one (float32, zlib-friendly: mostly small multiples of one ulp).  The tests replace their
    "measured + 10 %" constants by max(1e-4, 1.6 x this spread)."""
    from dmb.modeling.stereo.cost_processors import build_cost_processor
    from dmb.modeling.stereo.disp_predictors import build_disp_predictor
    from densematchingbenchmark_amd import synthetic

    out = {}
    threads = (8, 3, 1)
    with torch.no_grad():
        for tag, rel, (fh, fw), gain in (("s544", "configs/PSMNet/scene_flow.py", (136, 240), 10.0),
                                         ("kitti", "configs/PSMNet/kitti_2015.py", (96, 312), 10.0),
                                         ("g30", "configs/PSMNet/scene_flow.py", (136, 240), 30.0)):   # the gain-30 family of round 3
            cfg = G.load_cfg(rel)
            m = _M()
            m.cost_processor = build_cost_processor(cfg)
            m.disp_predictor = build_disp_predictor(cfg)
            m.eval()
            synthetic.init_params_(m, seed=0, classif_gain=gain)
            lf, rf = synthetic.feature_pair(0, 32, fh, fw)
            maps = {}
            for t in threads:
                torch.set_num_threads(t)
                costs = m.cost_processor(lf, rf)
                maps[t] = [m.disp_predictor(c) for c in costs]
                del costs
                for lvl, d in enumerate(maps[t]):
                    out["%s_t%d_disp%d" % (tag, t, 3 - lvl)] = G.npy(d[SUB])
                print(tag, "threads", t, "disp3 range %.2f..%.2f" % (maps[t][0].min().item(), maps[t][0].max().item()), flush=True)
            for lvl in range(3):
                full = max((maps[a][lvl] - maps[b][lvl]).abs().max().item() for a in threads for b in threads if a < b)
                sub = max((maps[a][lvl][SUB] - maps[b][lvl][SUB]).abs().max().item() for a in threads for b in threads if a < b)
                out["%s_spread_full_disp%d" % (tag, 3 - lvl)] = np.float64(full)
                out["%s_spread_sub_disp%d" % (tag, 3 - lvl)] = np.float64(sub)
                print(tag, "level", 3 - lvl, "self-spread: whole map %.3e, sub-sample %.3e" % (full, sub), flush=True)
            if tag != "g30":
                for t in threads[1:]:
                    out["%s_t%d_minus_t8_disp3_full" % (tag, t)] = G.npy(maps[t][0] - maps[8][0])
    torch.set_num_threads(int(os.environ.get("DMB_THREADS", "8")))
    np.savez_compressed(os.path.join(OUT, "fullsize_psmnet_spread.npz"), **out)
    print("fullsize_psmnet_spread.npz %8.1f KB" % (os.path.getsize(os.path.join(OUT, "fullsize_psmnet_spread.npz")) / 1024))


def peaked():
    """Round 6: a D = 192 fixture whose cost distributions are PEAKED the way a trained network's are.  The
    weights are the reference's own modules briefly trained on exact-match feature pairs (oracle/train_peaked_reference.py ->
    tests/golden/psmnet_trained_weights.npz, FP16-rounded: inputs); here the REAL reference evaluates them (eval mode) at 544x960 on
    a banded exact-match pair.  Stored: the sub-sampled maps and cost rows of the three levels, the whole best-level map, how
    peaked the distributions are (E|k - disp|, the largest probability) and the reference's self-spread on THIS fixture (1 / 3 / 8
    host threads, as `spread`): north_star's 1e-4 was written for this regime, and the GPU test asserts it here."""
    from dmb.modeling.stereo.cost_processors import build_cost_processor
    from dmb.modeling.stereo.disp_predictors import build_disp_predictor
    from densematchingbenchmark_amd import synthetic

    w = np.load(os.path.join(OUT, "psmnet_trained_weights.npz"))
    with torch.no_grad():
        cfg = G.load_cfg("configs/PSMNet/scene_flow.py")
        m = _M()
        m.cost_processor = build_cost_processor(cfg)
        m.disp_predictor = build_disp_predictor(cfg)
        sd = {k: torch.from_numpy(w[k].astype(np.float32) if w[k].dtype == np.float16 else w[k]) for k in w.files}
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not unexpected and all(k.endswith("disp_regression.weight") for k in missing), (missing, unexpected)
        m.eval()
        lf, rf, gt = synthetic.banded_match_pair(7, 136, 240, 48, bands=6)
        out, maps = {}, {}
        for t in (8, 3, 1):
            torch.set_num_threads(t)
            costs = m.cost_processor(lf, rf)
            maps[t] = [m.disp_predictor(c) for c in costs]
            if t == 8:
                for lvl, (d, c) in enumerate(zip(maps[t], costs)):
                    out["disp%d" % (3 - lvl)] = G.npy(d[SUB])
                    out["cost%d_rows" % (3 - lvl)] = G.npy(c[CROWS])
                out["disp3_full"] = G.npy(maps[t][0])
                pr = torch.softmax(costs[0], 1)
                k = torch.arange(192.).view(1, -1, 1, 1)
                mask = gt > 0
                width = (pr * (k - maps[t][0]).abs()).sum(1, keepdim=True)      # per-pixel E|k - disp| of the best level
                out["stats"] = np.array([(pr * (k - maps[t][0]).abs()).sum(1).mean().item(), pr.max(1)[0].mean().item(),
                                         (maps[t][0][mask] - gt[mask]).abs().mean().item(), costs[0].min().item(), costs[0].max().item()])
                print("peaked: E|k - disp| mean %.2f px, largest probability mean %.3f, EPE %.3f px, cost range %.1f..%.1f" % tuple(out["stats"]), flush=True)
            del costs
        for lvl in range(3):
            full = max((maps[a][lvl] - maps[b][lvl]).abs().max().item() for a in (8, 3, 1) for b in (8, 3, 1) if a < b)
            out["spread_full_disp%d" % (3 - lvl)] = np.float64(full)
            print("peaked level", 3 - lvl, "self-spread of the reference: %.3e" % full, flush=True)
        # Where on the map is a distribution peaked?  The per-pixel E|k - disp| of the best level (FP16) and the per-pixel self-spread
        # (max over the three pairs of evaluations), so that the test can state the bound per class of pixels: confidently matched
        # ones (E|k - disp| <= 2 px) against ambiguous / unmatched ones (band edges, columns without a match).
        out["disp3_width_full"] = width.numpy().astype(np.float16)
        sp = torch.zeros_like(maps[8][0])
        for a, b in ((8, 3), (8, 1), (3, 1)):
            sp = torch.maximum(sp, (maps[a][0] - maps[b][0]).abs())
        out["disp3_self_spread_full"] = G.npy(sp)
        for lo, hi in ((0.0, 1.0), (1.0, 2.0), (2.0, 4.0), (4.0, 1e9)):
            sel = (width >= lo) & (width < hi)
            print("  pixels with %g <= E|k - disp| < %g: %5.1f %%, self-spread max %.3e" % (lo, hi, 100.0 * sel.float().mean().item(),
                                                                                       sp[sel].max().item() if sel.any() else 0.0), flush=True)
    torch.set_num_threads(int(os.environ.get("DMB_THREADS", "8")))
    np.savez_compressed(os.path.join(OUT, "fullsize_psmnet_peaked.npz"), **out)
    print("fullsize_psmnet_peaked.npz %8.1f KB" % (os.path.getsize(os.path.join(OUT, "fullsize_psmnet_peaked.npz")) / 1024))


def main():
    G.import_reference()
    torch.set_num_threads(int(os.environ.get("DMB_THREADS", "8")))
    if "round3" in sys.argv[1:]:
        round3()
        return
    if "kitti" in sys.argv[1:]:
        kitti()
        return
    if "spread" in sys.argv[1:]:
        spread()
        return
    if "peaked" in sys.argv[1:]:
        peaked()
        return
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
    round3()
    kitti()
    for f in sorted(os.listdir(OUT)):
        if f.startswith("fullsize"):
            print("%-28s %8.1f KB" % (f, os.path.getsize(os.path.join(OUT, f)) / 1024))


if __name__ == "__main__":
    main()
