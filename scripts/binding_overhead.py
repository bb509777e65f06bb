"""Round 6: what the host pays per launch through the Python wrapper + ctypes binding.
  (a) host microseconds per ops.conv3d_k3 call with the GPU kept far behind (a tiny launch, the queue never drains into the timing);
      split into: the bare ctypes call, pointer / stream marshalling, output allocation, the module layer above it;
  (b) one BASELINE configs[0] step (one 256x512 pair, 46 launches): eager wall time per step against the same step replayed from a
      HIP graph, and a cProfile of the eager step's host side;
  (c) eager steps whose shapes VARY from call to call (a graph does not help there)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from densematchingbenchmark_amd import _lib, ops, synthetic
from densematchingbenchmark_amd.config import Config
from densematchingbenchmark_amd.modeling import build_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
lib = _lib.load()


def host_us(fn, n=2000):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt / n * 1e6


# (a) one tiny launch: [1, 16, 2, 4, 16] -> 32 channels
x = torch.randn(1, 16, 2, 4, 16, device=dev)
wp = ops.pack_conv3d_weights(torch.randn(32, 16, 3, 3, 3, device=dev))
sc, sh = torch.ones(32, device=dev), torch.zeros(32, device=dev)
y = torch.empty(1, 32, 2, 4, 16, device=dev)
px, pw, ps, ph, py = (_lib.dev_ptr(t) for t in (x, wp, sc, sh, y))
st = _lib.stream_ptr(dev)
print("(a) host microseconds per call, tiny launch (the device never becomes the bottleneck: %d-thread host)" % (os.cpu_count() or 0))
print("    bare ctypes call, arguments prepared:        %6.2f us" % host_us(lambda: lib.dmb_conv3d_k3_f32(px, pw, ps, ph, None, py, 1, 16, 32, 2, 4, 16, 1, 1, st)))
print("    + dev_ptr x 5 + stream_ptr + check:          %6.2f us" % host_us(lambda: _lib.check(lib.dmb_conv3d_k3_f32(
    _lib.dev_ptr(x), _lib.dev_ptr(wp), _lib.dev_ptr(sc), _lib.dev_ptr(sh), None, _lib.dev_ptr(y), 1, 16, 32, 2, 4, 16, 1, 1, _lib.stream_ptr(dev)), "c")))
print("    stream_ptr alone:                            %6.2f us" % host_us(lambda: _lib.stream_ptr(dev)))
print("    torch.empty of the output alone:             %6.2f us" % host_us(lambda: torch.empty(1, 32, 2, 4, 16, device=dev)))
shim = _lib.shim()
print("    torch-extension shim: %s" % _lib.shim_state())


def binding(which):
    """Switch ops between the two bindings of the same C ABI (the shim when it is loaded, else ctypes)."""
    _lib._shim = shim if which == "shim" else None


for which in (("ctypes", "shim") if shim is not None else ("ctypes",)):
    binding(which)
    print("    ops.conv3d_k3(out=y) through %-7s          %6.2f us" % (which + ":", host_us(lambda: ops.conv3d_k3(x, wp, 32, sc, sh, None, 1, True, out=y))))
    print("    ops.conv3d_k3 (allocates) through %-7s     %6.2f us" % (which + ":", host_us(lambda: ops.conv3d_k3(x, wp, 32, sc, sh, None, 1, True))))
if shim is not None:
    print("    the shim's function called directly:         %6.2f us" % host_us(lambda: shim.conv3d_k3(x, wp, 32, sc, sh, None, 1, 1, y)))

# (b) one configs[0] step
cfg = Config.fromfile(os.path.join(ROOT, "configs", "PSMNet", "baseline_cfg0_256x512_d64.py"))
model = build_model(cfg, backbone=None).eval()
synthetic.init_params_(model, seed=0, classif_gain=10.0)
model = model.to(dev)
ops.set_branch_overlap(False)
lf, rf = synthetic.feature_batch(0, 1, 1, 32, 64, 128, dev)
batch = dict(leftFeature=lf, rightFeature=rf)


def step(b=batch):
    with torch.no_grad():
        return model(b)[0]["disps"]


def wall_ms(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def host_only_ms(fn, n=200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt / n * 1e3


from densematchingbenchmark_amd.graph_runner import GraphedForward
g = GraphedForward(model, track_parameters=False)
with torch.no_grad():
    g(batch)
print("(b) BASELINE configs[0], one 256x512 pair, the path alone (46 launches):")
for which in (("ctypes", "shim") if shim is not None else ("ctypes",)):
    binding(which)
    print("    eager through %-7s back to back: %6.3f ms per step (wall)   host issue time alone %6.3f ms" % (which + ",", wall_ms(step), host_only_ms(step)))
print("    HIP-graph replay:                     %6.3f ms per step" % wall_ms(lambda: g(batch)))
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
torch.cuda.synchronize()
s = pstats.Stats(pr)
s.sort_stats("tottime")
print("    cProfile of 200 eager steps, by own time:")
s.print_stats(18)

# (c) varying shapes: 5 widths in turn
shapes = [(64, 128), (64, 112), (56, 128), (64, 96), (48, 128)]
batches = [dict(zip(("leftFeature", "rightFeature"), synthetic.feature_batch(i, 1, 1, 32, h, w, dev))) for i, (h, w) in enumerate(shapes)]
k = [0]


def varying():
    k[0] += 1
    return step(batches[k[0] % len(batches)])


gv = GraphedForward(model, track_parameters=False, max_graphs=8)


def varying_graph():
    k[0] += 1
    with torch.no_grad():
        return gv(batches[k[0] % len(batches)])


print("(c) five feature shapes in turn (%s):" % ", ".join("%dx%d" % s_ for s_ in shapes))
print("    eager:                         %6.3f ms per step" % wall_ms(varying))
print("    one graph per shape (5 held):  %6.3f ms per step" % wall_ms(varying_graph))
