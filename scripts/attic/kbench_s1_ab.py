"""Development aid: the dominant stride-1 layer (32 -> 32, full resolution, batch 4) and its KITTI sibling under ONE build of the
library, a checksum of the output included -- run once per build-time variant (DMB_BUILD_TAG=... DMB_BUILD_DEFS=... python -m
densematchingbenchmark_amd.build --dev; DMB_LIB=dev_<tag>) to A/B variants that cannot live in one process."""
import os, sys
os.environ.setdefault("DMB_LIB", "dev")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from densematchingbenchmark_amd import _lib, ops
dev = torch.device("cuda:0"); lib = _lib.load()


def timeit(fn, n=30, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


torch.manual_seed(0)
for (B, D, H, W) in ((4, 48, 136, 240), (4, 48, 96, 312)):
    x = torch.randn(B, 32, D, H, W, device=dev)
    wp = ops.pack_conv3d_weights(torch.randn(32, 32, 3, 3, 3, device=dev) * 0.03)
    sc, sh = torch.ones(32, device=dev), torch.zeros(32, device=dev)
    res = torch.randn(B, 32, D, H, W, device=dev)
    y = ops.conv3d_k3(x, wp, 32, sc, sh, res, 1, True)
    ts = [timeit(lambda: ops.conv3d_k3(x, wp, 32, sc, sh, None, 1, True)) for _ in range(3)]
    tr = [timeit(lambda: ops.conv3d_k3(x, wp, 32, sc, sh, res, 1, True)) for _ in range(3)]
    fl = 2.0 * 27 * 32 * 32 * B * D * H * W
    print("%s [%d,32,%d,%d,%d]: %.4f ms (%.3f of peak)   with skip %.4f ms   checksum %.6f" %
          (os.environ["DMB_LIB"], B, D, H, W, min(ts), fl / min(ts) / 1e9 / 157.3, min(tr), y.double().sum().item()), flush=True)
