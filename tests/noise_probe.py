"""FP32 noise floor of the path: GPU (HIP) and CPU oracle, each against an FP64 evaluation of the same network."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densematchingbenchmark_amd import synthetic
from densematchingbenchmark_amd.config import Config
from densematchingbenchmark_amd.modeling import build_model
from oracle import dmb_oracle as O

gain = float(os.environ.get("GAIN", "30"))
md = int(os.environ.get("MD", "64"))
fh, fw = int(os.environ.get("FH", "64")), int(os.environ.get("FW", "128"))
cfg = Config.fromfile("configs/PSMNet/scene_flow.py")
cfg.model.max_disp = md
cfg.model.cost_processor.cost_computation.max_disp = md // 4
cfg.model.cost_processor.cost_aggregator.max_disp = md
cfg.model.disp_predictor.max_disp = md
model = build_model(cfg, backbone=None).eval()
synthetic.init_params_(model, seed=0, classif_gain=gain)
p = {k: v.clone() for k, v in model.state_dict().items()}
lf, rf = synthetic.feature_pair(0, 32, fh, fw)
torch.set_num_threads(32)
with torch.no_grad():
    d32, c32 = O.psmnet_path(lf, rf, p, md)
    q32 = O.psm_aggregator(O.cat_fms(lf, rf, md // 4, 0, 1), p, md, "cost_processor.aggregator.", upsample=False)
    p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in p.items()}
    raw = O.cat_fms(lf, rf, md // 4, 0, 1).double()
    c64 = O.psm_aggregator(raw, p64, md, "cost_processor.aggregator.")
    q64 = O.psm_aggregator(raw, p64, md, "cost_processor.aggregator.", upsample=False)
    d64 = [O.soft_argmin_f64(c, md) for c in c64]
    dev = torch.device("cuda:0")
    m = model.to(dev)
    res, _ = m(dict(leftFeature=lf.to(dev), rightFeature=rf.to(dev)))
    agg = m.cost_processor.aggregator
    qg = agg.trunk(m.cost_processor.vol_func(lf.to(dev), rf.to(dev), **m.cost_processor.default_args))
print("gain", gain, "md", md, "feat", fh, fw, "cost range %.2f..%.2f" % (c64[0].min().item(), c64[0].max().item()))
for i in range(3):
    qgpu = qg[2 - i].squeeze(1).cpu().double()
    print("level %d  quarter-res cost err vs fp64: gpu %.2e  oracle32 %.2e | full cost: gpu %.2e oracle32 %.2e | disp: gpu %.2e (mean %.2e) oracle32 %.2e (mean %.2e) | gpu-vs-oracle32 disp %.2e" % (
        3 - i, (qgpu - q64[i]).abs().max().item(), (q32[i].double() - q64[i]).abs().max().item(),
        (res["costs"][i].cpu().double() - c64[i]).abs().max().item(), (c32[i].double() - c64[i]).abs().max().item(),
        (res["disps"][i].cpu().double() - d64[i]).abs().max().item(), (res["disps"][i].cpu().double() - d64[i]).abs().mean().item(),
        (d32[i].double() - d64[i]).abs().max().item(), (d32[i].double() - d64[i]).abs().mean().item(),
        (res["disps"][i].cpu() - d32[i]).abs().max().item()))
