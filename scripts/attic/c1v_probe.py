"""Development aid: the 32 -> 1 head convolution (conv3d_c1v_kernel) at the BASELINE shape, with and without the skip operand, and its
checksum (the kernel's summation order is fixed: any restructuring has to reproduce the bits).   python scripts/attic/c1v_probe.py"""
import os, sys
os.environ.setdefault("DMB_LIB", "dev")   # kernel-variant switches exist only in the development build (build.py --dev)
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from densematchingbenchmark_amd import _lib, ops
dev = torch.device("cuda:0"); lib = _lib.load()
def timeit(fn, n=30, warm=6):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
g = torch.Generator(device="cpu").manual_seed(5)
x = torch.randn(4, 32, 48, 136, 240, generator=g).to(dev)
r = torch.randn(4, 1, 48, 136, 240, generator=g).to(dev)
w = (torch.randn(1, 32, 3, 3, 3, generator=g) * 0.05).to(dev)
y = ops.conv3d_k3_c1(x, w, 0.25, r)
print("checksum %.9e  |y|max %.6f" % (y.double().sum().item(), y.abs().max().item()))
lib.dmb_dev_set_option(3, 1)   # scalar kernel
y0 = ops.conv3d_k3_c1(x, w, 0.25, r)
lib.dmb_dev_set_option(3, 0)
print("bit-identical to conv3d_c1_kernel:", bool(torch.equal(y, y0)))
for rep in range(3):
    print("no operand %.4f ms   with operand %.4f ms" % (timeit(lambda: ops.conv3d_k3_c1(x, w, 0.25, None)),
                                                         timeit(lambda: ops.conv3d_k3_c1(x, w, 0.25, r))), flush=True)
