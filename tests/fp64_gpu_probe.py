"""Development aid: can the oracle's FP64 evaluation (the yardstick of tests/test_fullsize_gpu.py) run on the GPU through torch's own
double-precision kernels, and how long does it take there vs on the host?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densematchingbenchmark_amd import synthetic
from densematchingbenchmark_amd.config import Config
from densematchingbenchmark_amd.modeling import build_model
from oracle import dmb_oracle as O

cfg = Config.fromfile("configs/PSMNet/scene_flow.py")
model = build_model(cfg, backbone=None).eval()
synthetic.init_params_(model, seed=0, classif_gain=10.0)
p = {k: v.clone() for k, v in model.state_dict().items()}
lf, rf = synthetic.feature_pair(0, 32, 136, 240)
dev = torch.device("cuda:0")
p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in p.items()}
with torch.no_grad():
    raw = O.cat_fms(lf, rf, 48, 0, 1).double()
    for threads in (32, 128):
        torch.set_num_threads(threads)
        t0 = time.time()
        c_cpu = O.psm_aggregator(raw, p64, 192, "cost_processor.aggregator.")
        print("cpu fp64 aggregator, %d threads: %.1f s" % (threads, time.time() - t0), flush=True)
    try:
        pg = {k: v.to(dev) for k, v in p64.items()}
        rg = raw.to(dev)
        torch.cuda.synchronize()
        t0 = time.time()
        c_gpu = O.psm_aggregator(rg, pg, 192, "cost_processor.aggregator.")
        torch.cuda.synchronize()
        print("gpu fp64 aggregator (torch kernels): %.1f s" % (time.time() - t0), flush=True)
        for a, b in zip(c_cpu, c_gpu):
            print("  cpu-vs-gpu fp64 max diff %.3g" % (a - b.cpu()).abs().max().item())
    except Exception as e:  # noqa: BLE001
        print("gpu fp64 failed:", repr(e)[:300])
