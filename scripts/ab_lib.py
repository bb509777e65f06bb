"""Development aid: time a kernel CHANGE against the previous build of the library inside ONE process on ONE chip (numbers from
different gpurun boxes differ by 1-2 %, more than most changes are worth).
    cp densematchingbenchmark_amd/lib/libdmb_hip.so densematchingbenchmark_amd/lib/libdmb_hip_alt.so     # the build to compare with
    ... edit, rebuild ...
    python scripts/ab_lib.py deconv6 deconv5 s2_1 s1f          # cases as in scripts/kcase.py (+ "step": the whole PSMNet step)
Both libraries are linked -Bsymbolic (build.py), so each one's entry points call its own kernels; the outputs of the two builds are
compared bit for bit as well."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from densematchingbenchmark_amd import _lib, ops

dev = torch.device("cuda:0")
new = _lib.load()
alt_path = os.path.join(ROOT, "densematchingbenchmark_amd", "lib", "libdmb_hip_alt.so")
alt = ctypes.CDLL(alt_path, mode=ctypes.RTLD_LOCAL)
for name, (res, args) in _lib.SIGNATURES.items():
    if hasattr(alt, name):
        fn = getattr(alt, name)
        fn.restype, fn.argtypes = res, args


def use(lib):
    _lib._lib = lib


B, D, H, W = 4, 48, 136, 240
g = lambda *s: torch.randn(*s, device=dev)


def make(name):
    if name in ("deconv6", "deconv5"):
        Ci, Co, d, h, w = (64, 32, D // 2, H // 2, W // 2) if name == "deconv6" else (64, 64, D // 4, H // 4, W // 4)
        x, wt = g(B, Ci, d, h, w), g(Ci, Co, 3, 3, 3) * 0.03
        sc, sh, r = torch.ones(Co, device=dev), torch.zeros(Co, device=dev), g(B, Co, 2 * d, 2 * h, 2 * w)
        wp = ops.pack_deconv3d_weights(wt)   # (the packed layout is the same in both builds)
        return lambda: ops.deconv3d_k3s2(x, wp, Co, sc, sh, r, True), True
    if name in ("s2_1", "s2_3", "s1q", "s1f", "s1h", "s1fr", "s1hr"):
        Ci, Co, st, d, h, w = {"s2_1": (32, 64, 2, D, H, W), "s2_3": (64, 64, 2, D // 2, H // 2, W // 2),
                               "s1q": (64, 64, 1, D // 4, H // 4, W // 4), "s1f": (32, 32, 1, D, H, W), "s1fr": (32, 32, 1, D, H, W),
                               "s1h": (64, 64, 1, D // 2, H // 2, W // 2), "s1hr": (64, 64, 1, D // 2, H // 2, W // 2)}[name]
        x, wt = g(B, Ci, d, h, w), g(Co, Ci, 3, 3, 3) * 0.03
        sc, sh = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
        r = g(B, Co, (d - 1) // st + 1, (h - 1) // st + 1, (w - 1) // st + 1) if name.endswith("r") else None
        wp = ops.pack_conv3d_weights(wt)
        return lambda: ops.conv3d_k3(x, wp, Co, sc, sh, r, st, True), True
    if name == "c1":
        x, w1, r = g(B, 32, D, H, W), g(1, 32, 3, 3, 3), g(B, 1, D, H, W)
        return lambda: ops.conv3d_k3_c1(x, w1, 0.0, r), True
    if name == "step":
        from densematchingbenchmark_amd import synthetic
        from densematchingbenchmark_amd.config import Config
        from densematchingbenchmark_amd.modeling import build_model
        cfg = Config.fromfile(os.path.join(ROOT, "configs", "PSMNet", "scene_flow.py"))
        model = build_model(cfg).eval()
        synthetic.init_params_(model, seed=0, classif_gain=10.0)
        model = model.to(dev)
        left, right = synthetic.feature_batch(0, 1, 4, 32, 136, 240, dev)
        batch = dict(leftFeature=left, rightFeature=right)
        return lambda: model(batch)[0]["disps"][0], False
    raise SystemExit("unknown case " + name)


def timeit(fn, n, warm):
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


with torch.no_grad():
    for name in sys.argv[1:] or ["deconv6"]:
        fn, small = make(name)
        n, warm = (20, 5) if small else (8, 2)
        outs, times = {}, {"new": [], "previous": []}
        for rep in range(5):   # (the first repetition warms up and is dropped)
            for tag, lib in (("new", new), ("previous", alt)):
                use(lib)
                times[tag].append(timeit(fn, n, warm))
                outs[tag] = fn().clone()
        use(new)
        same = bool(torch.equal(outs["new"], outs["previous"]))
        med = {t: sorted(v[1:])[len(v[1:]) // 2] for t, v in times.items()}
        print("%-8s new %.4f ms   previous %.4f ms   (%+.2f %%)   outputs bit-identical: %s   [%s | %s]" % (
            name, med["new"], med["previous"], 100 * (med["new"] / med["previous"] - 1), same,
            " ".join("%.4f" % v for v in times["new"]), " ".join("%.4f" % v for v in times["previous"])), flush=True)
