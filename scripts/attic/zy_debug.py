"""Development aid for csrc/deconv3d_zy.hip: compare the (tile, z parity, y parity) form with deconv3d_kernel on one shape,
optionally with a forced small grid (few workgroups walk many items: cross-item pipelining and class switches), and say WHERE
they differ.   python scripts/attic/zy_debug.py Ci Co B D H W [grid]"""
import os
os.environ.setdefault("DMB_LIB", "dev")   # kernel-variant switches exist only in the development build (build.py --dev)
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from densematchingbenchmark_amd import _lib, ops

Ci, Co, B, D, H, W = [int(v) for v in sys.argv[1:7]]
grid = int(sys.argv[7]) if len(sys.argv) > 7 else 0
dbg = int(sys.argv[8]) if len(sys.argv) > 8 else 0
dev = torch.device("cuda:0")
lib = _lib.load()
g = torch.Generator().manual_seed(1)
x = torch.randn((B, Ci, D, H, W), generator=g).to(dev)
w = (torch.randn((Ci, Co, 3, 3, 3), generator=g) * 0.05).to(dev)
wp = ops.pack_deconv3d_weights(w)
lib.dmb_dev_set_option(4, 1)
ref = ops.deconv3d_k3s2(x, wp, Co, None, None, None, False)
lib.dmb_dev_set_option(4, 0)
lib.dmb_dev_set_option(9, grid)
lib.dmb_dev_set_option(6, dbg)
lib.dmb_dev_set_option(16, int(os.environ.get("RUN", "0")))    # RUN=R: item order in groups of R tiles (0 = the default)
lib.dmb_dev_set_option(20, int(os.environ.get("ONE", "0")))    # ONE=1: one class-major list over the whole layer (round-3 order)
got = ops.deconv3d_k3s2(x, wp, Co, None, None, None, False)
lib.dmb_dev_set_option(9, 0)
lib.dmb_dev_set_option(6, 0)
lib.dmb_dev_set_option(16, 0)
lib.dmb_dev_set_option(20, 0)
torch.cuda.synchronize()
bad = (got != ref)
print("shape", (Ci, Co, B, D, H, W), "grid", grid, "dbg", dbg, "mismatching elements:", int(bad.sum()), "of", bad.numel(), "max diff", float((got - ref).abs().max()))
if bad.any():
    for pz in (0, 1):
        for py in (0, 1):
            sub = bad[:, :, pz::2, py::2, :]
            print("  class (pz=%d, py=%d): %d bad" % (pz, py, int(sub.sum())), end="")
            if sub.any():
                idx = sub.nonzero()
                print("  first:", idx[0].tolist(), " last:", idx[-1].tolist(), " batch items:", sorted(set(idx[:, 0].tolist())),
                      " channels: %d..%d" % (int(idx[:, 1].min()), int(idx[:, 1].max())),
                      " z(in): %s" % sorted(set(idx[:, 2].tolist()))[:12], " y(in): %s" % sorted(set(idx[:, 3].tolist()))[:12],
                      " x(out): %d..%d" % (int(idx[:, 4].min()), int(idx[:, 4].max())))
            else:
                print()
if bad.any():
    rows = bad.any(dim=4).nonzero()
    print("bad rows:", rows.shape[0], "of", bad.shape[0] * bad.shape[1] * bad.shape[2] * bad.shape[3])
    import collections
    hist = collections.Counter()
    for r in rows[:2000].tolist():
        xs = bad[r[0], r[1], r[2], r[3]].nonzero().flatten().tolist()
        hist[(xs[0], xs[-1], len(xs))] += 1
    print("  (first bad x, last bad x, count) histogram over the first 2000 bad rows:", hist.most_common(12))
    chist = collections.Counter(rows[:, 1].tolist())
    print("  bad rows per channel:", sorted(chist.items())[:40])
    for r in rows[:3].tolist():
        b, c, z, y = r
        xs = bad[b, c, z, y].nonzero().flatten()
        print("  row", r, "bad x:", xs.tolist()[:8], "... got", got[b, c, z, y, xs[:4]].tolist(), "ref", ref[b, c, z, y, xs[:4]].tolist())
        # is the wrong data some other row's correct data?
        g4 = got[b, c, z, y, xs[0]:xs[0] + 4]
        hit = ((ref[b, :, :, :, xs[0]:xs[0] + 4] - g4).abs().sum(-1) == 0).nonzero()
        print("     the same 4 values appear in ref at (channel, z, y):", hit[:6].tolist())
