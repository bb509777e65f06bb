"""StereoNet-8x end to end at KITTI size (BASELINE configs[4] shape: 1242x375 padded to 1248x384): stage times
(development aid; bench.py is the judged entry)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densematchingbenchmark_amd import synthetic
from densematchingbenchmark_amd.config import Config
from densematchingbenchmark_amd.modeling import build_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
B = int(os.environ.get("KB_B", "16"))
H, W = 384, 1248


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


cfg = Config.fromfile(os.path.join(ROOT, "configs", "StereoNet", "scene_flow_8x_2stage.py"))
cfg.model.backbone = dict(type="StereoNet", in_planes=3, downsample_num=3, residual_num=6)
cfg.model.disp_refinement = dict(type="StereoNet", in_planes=4, num=1)
model = build_model(cfg, backbone="hip").eval()
synthetic.init_params_(model, seed=12, classif_gain=10.0)
model = model.to(dev)
li, ri = torch.randn(B, 3, H, W, device=dev), torch.randn(B, 3, H, W, device=dev)
with torch.no_grad():
    lf, rf = model.backbone(li, ri)
    t_bb = timeit(lambda: model.backbone(li, ri))
    costs = model.cost_processor(lf, rf)
    t_cp = timeit(lambda: model.cost_processor(lf, rf))
    disps = [model.disp_predictor(c) for c in costs]
    t_dp = timeit(lambda: [model.disp_predictor(c) for c in costs])
    t_rf = timeit(lambda: model.disp_refinement(disps, lf, rf, li, ri))
    t_all = timeit(lambda: model(dict(leftImage=li, rightImage=ri)))
gf_bb = 2 * B * (2 * 25 * 3 * 32 * (H // 2) * (W // 2) + 2 * 25 * 32 * 32 * ((H // 4) * (W // 4) + (H // 8) * (W // 8))
                 + 13 * 2 * 9 * 32 * 32 * (H // 8) * (W // 8)) / 1e9
gf_rf = B * (2 * 9 * H * W * (4 * 32 + 12 * 32 * 32 + 32)) / 1e9
print("StereoNet-8x, %d pairs of %dx%d" % (B, H, W))
print("backbone     %8.3f ms  (%.1f GFLOP -> %.1f TFLOP/s)" % (t_bb, gf_bb, gf_bb / t_bb))
print("cost path    %8.3f ms  (volume + aggregator) + %.3f ms soft-argmin" % (t_cp, t_dp))
print("refinement   %8.3f ms  (%.1f GFLOP -> %.1f TFLOP/s)" % (t_rf, gf_rf, gf_rf / t_rf))
print("end to end   %8.3f ms  -> %.1f pairs/s" % (t_all, B / t_all * 1e3))
