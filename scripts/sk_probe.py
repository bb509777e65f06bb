"""Development aid (round 6): the split-K convolution (csrc/conv3d_sk.hip) and the split-channel head (conv3d_c1s_kernel) --
correctness of every variant against torch CPU on awkward shapes, then time per variant against the full-grid kernels at the
hourglass shapes of one pair (BASELINE configs[0], 544x960, KITTI).  Development library (options 23 / 24).  KB_B = batch."""
import os
os.environ.setdefault("DMB_LIB", "dev")
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from densematchingbenchmark_amd import _lib, ops

dev = torch.device("cuda:0")
B = int(os.environ.get("KB_B", "1"))
lib = _lib.load()


def timeit(fn, n=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def timeit_graph(fn, n=20, reps=10):
    """GPU time per launch without the host: n launches captured in a HIP graph, replayed."""
    fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        fn()
        st.synchronize()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (n * reps) * 1e3


def check():
    g = torch.Generator().manual_seed(5)
    bad = 0
    for (b, Ci, Co, D, H, W, stride) in ((1, 64, 64, 4, 16, 32, 1), (2, 32, 64, 3, 7, 20, 1), (1, 16, 32, 5, 9, 36, 1), (1, 64, 64, 8, 32, 64, 2),
                                         (2, 32, 64, 5, 11, 44, 2), (1, 48, 32, 3, 6, 28, 2), (1, 64, 64, 2, 5, 12, 1), (1, 32, 32, 6, 10, 52, 2)):
        x = torch.randn((b, Ci, D, H, W), generator=g)
        w = torch.randn((Co, Ci, 3, 3, 3), generator=g) / (Ci * 27) ** 0.5
        sc, sh = 0.5 + torch.rand(Co, generator=g), torch.rand(Co, generator=g) - 0.5
        ref0 = F.conv3d(x, w, None, stride=stride, padding=1) * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1)
        res = torch.randn(ref0.shape, generator=g)
        wp = ops.pack_conv3d_weights(w.to(dev))
        for relu, use_res in ((True, True), (False, False), ("pre", True)):
            ref = ref0
            if relu == "pre":
                ref = F.relu(ref)
            if use_res:
                ref = ref + res
            if relu is True:
                ref = F.relu(ref)
            for v in (0, 1, 2, 3):
                lib.dmb_dev_set_option(23, 1 if v == 0 else v + 1)
                got = ops.conv3d_k3(x.to(dev), wp, Co, sc.to(dev), sh.to(dev), res.to(dev) if use_res else None, stride, relu)
                err = (got.cpu() - ref).abs().max().item()
                if not err <= 3e-5:
                    bad += 1
                    print("FAIL variant", v, (b, Ci, Co, D, H, W, stride, relu, use_res), err)
    lib.dmb_dev_set_option(23, 0)
    for (b, Ci, Co, D, H, W) in ((1, 64, 64, 4, 16, 32), (2, 64, 32, 3, 5, 20), (1, 32, 64, 2, 7, 36), (1, 16, 32, 5, 3, 12), (1, 64, 1, 2, 4, 16)):
        x = torch.randn((b, Ci, D, H, W), generator=g)
        w = torch.randn((Ci, Co, 3, 3, 3), generator=g) / (Ci * 27 / 8) ** 0.5
        sc, sh = 0.5 + torch.rand(Co, generator=g), torch.rand(Co, generator=g) - 0.5
        ref0 = F.conv_transpose3d(x, w, None, stride=2, padding=1, output_padding=1) * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1)
        res = torch.randn(ref0.shape, generator=g)
        wp = ops.pack_deconv3d_weights(w.to(dev))
        for relu, use_res in ((True, True), (False, False), ("pre", True)):
            ref = ref0
            if relu == "pre":
                ref = F.relu(ref)
            if use_res:
                ref = ref + res
            if relu is True:
                ref = F.relu(ref)
            for v in (0, 1, 2, 3, 4):
                lib.dmb_dev_set_option(25, 1 if v == 0 else v + 1)
                got = ops.deconv3d_k3s2(x.to(dev), wp, Co, sc.to(dev), sh.to(dev), res.to(dev) if use_res else None, relu)
                err = (got.cpu() - ref).abs().max().item()
                if not err <= 3e-5:
                    bad += 1
                    print("FAIL transposed variant", v, (b, Ci, Co, D, H, W, relu, use_res), err)
    lib.dmb_dev_set_option(25, 0)
    for (b, Ci, D, H, W) in ((1, 32, 16, 64, 128), (2, 5, 3, 9, 60), (1, 32, 7, 13, 124), (1, 2, 1, 1, 4), (1, 33, 5, 17, 64)):
        x, w = torch.randn((b, Ci, D, H, W), generator=g), torch.randn((1, Ci, 3, 3, 3), generator=g) / (Ci * 27) ** 0.5
        res = torch.randn((b, 1, D, H, W), generator=g)
        ref = F.conv3d(x, w, None, padding=1) + 0.25 + res
        outs = []
        for o in (1, 2):
            lib.dmb_dev_set_option(24, o)
            got = ops.conv3d_k3_c1(x.to(dev), w.to(dev), 0.25, res.to(dev))
            outs.append(got)
            err = (got.cpu() - ref).abs().max().item()
            if not err <= 3e-5:
                bad += 1
                print("FAIL head form", o, (b, Ci, D, H, W), err)
        print("head", (b, Ci, D, H, W), "split-channel vs single-chain max diff %.2e" % (outs[0] - outs[1]).abs().max().item())
    lib.dmb_dev_set_option(24, 0)
    print("correctness: failures", bad, flush=True)


def conv(Ci, Co, stride, d, h, w, res=False):
    x = torch.randn(B, Ci, d, h, w, device=dev)
    wp = ops.pack_conv3d_weights(torch.randn(Co, Ci, 3, 3, 3, device=dev) * 0.03)
    sc, sh = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
    do, ho, wo = [(n - 1) // stride + 1 for n in (d, h, w)]
    r = torch.randn(B, Co, do, ho, wo, device=dev) if res else None
    fl = 2.0 * 27 * Ci * Co * B * do * ho * wo
    out = torch.empty(B, Co, do, ho, wo, device=dev)
    return (lambda: ops.conv3d_k3(x, wp, Co, sc, sh, r, stride, True, out=out)), fl


def line(name, us, fl, ug=None):
    print("  %-58s %8.1f us  %6.1f TFLOP/s (%.2f of peak)%s" % (name, us, fl / us / 1e6, fl / us / 1e6 / 157.3,
                                                              "   graph replay %6.1f us" % ug if ug is not None else ""), flush=True)


check()
SHAPES = [tuple(int(v) for v in t.split("x")) for t in os.environ.get("KB_SHAPES", "16x64x128,48x136x240,48x96x312").split(",")]
for D, H, W in SHAPES:
    print("B = %d, quarter-resolution volume %d x %d x %d" % (B, D, H, W))
    for name, (Ci, Co, s, sc, res) in (("dres 32->32 full", (32, 32, 1, 1, False)), ("conv1 s2 32->64 full->half", (32, 64, 2, 1, False)),
                                       ("conv2 s1 64->64 half", (64, 64, 1, 2, True)), ("conv3 s2 64->64 half->quarter", (64, 64, 2, 2, False)),
                                       ("conv4 s1 64->64 quarter", (64, 64, 1, 4, False))):
        w_ = W // sc
        if w_ % 4:
            w_ = (w_ + 3) // 4 * 4
        f, fl = conv(Ci, Co, s, D // sc, H // sc, w_, res)
        for v, what in ((0, "full-grid kernel"), (1, "split-K 16x2, one row tile"), (2, "split-K 32x2, one row tile"), (3, "split-K 32x2, both row tiles")):
            lib.dmb_dev_set_option(23, 1 if v == 0 else v + 1)
            try:
                line(name + ": " + what, timeit(f), fl, timeit_graph(f))
            except Exception as e:  # noqa: BLE001
                print("  %s %s: %r" % (name, what, e))
        lib.dmb_dev_set_option(23, 1)
        if Co == 32 and s == 1:
            for c, what in ((3, "32x4 tile"), (4, "32x2 tile"), (8, "32x4 tile, chunks of 4"), (9, "32x2 tile, chunks of 4")):
                lib.dmb_dev_set_option(19, c)
                line(name + ": full-grid " + what, timeit(f), fl, timeit_graph(f))
            lib.dmb_dev_set_option(19, 0)
        if Co == 64:   # the full-grid kernels' small tiles with longer chunks
            opt, cands = (19, ((6, "16x2 tile, chunks of 2 (round 5)"), (7, "16x2 tile, chunks of 8"))) if s == 1 else (10, ((5, "one-row tile, chunks of 4"), (6, "two-row tile, chunks of 4")))
            for c, what in cands:
                lib.dmb_dev_set_option(opt, c)
                line(name + ": full-grid " + what, timeit(f), fl, timeit_graph(f))
            lib.dmb_dev_set_option(opt, 0)
        lib.dmb_dev_set_option(23, 0)
    for name, (Ci, Co, sc) in (("conv5 deconv 64->64 quarter->half +res", (64, 64, 4)), ("conv6 deconv 64->32 half->full +res", (64, 32, 2))):
        w_ = W // sc
        if w_ % 4:
            w_ = (w_ + 3) // 4 * 4
        xd = torch.randn(B, Ci, D // sc, H // sc, w_, device=dev)
        wpd = ops.pack_deconv3d_weights(torch.randn(Ci, Co, 3, 3, 3, device=dev) * 0.03)
        s1, s0 = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
        rd = torch.randn(B, Co, 2 * (D // sc), 2 * (H // sc), 2 * w_, device=dev)
        od = torch.empty_like(rd)
        fd = lambda: ops.deconv3d_k3s2(xd, wpd, Co, s1, s0, rd, True, out=od)   # noqa: E731
        fl = 2.0 * 27 * Ci * Co * xd[:, 0].numel()
        for v, what in ((0, "full-grid kernel"), (1, "split-K 16x2, 8 waves"), (2, "split-K 32x2, 8 waves"), (3, "split-K 16x2, 4 waves"), (4, "split-K 32x2, 4 waves")):
            lib.dmb_dev_set_option(25, 1 if v == 0 else v + 1)
            line(name + ": " + what, timeit(fd), fl, timeit_graph(fd))
        lib.dmb_dev_set_option(25, 0)
    x = torch.randn(B, 32, D, H, W, device=dev)
    wc = torch.randn(1, 32, 3, 3, 3, device=dev) * 0.03
    for o, what in ((1, "single chain (c1v)"), (2, "split channels (c1s)")):
        lib.dmb_dev_set_option(24, o)
        yo = torch.empty(B, 1, D, H, W, device=dev)
        ws = wc.reshape(-1).contiguous()
        hf = lambda: _lib.check(lib.dmb_conv3d_k3_c1_f32(_lib.dev_ptr(x), _lib.dev_ptr(ws), 0.0, None, _lib.dev_ptr(yo), B, 32, D, H, W, 0, _lib.stream_ptr(dev)), "c1")  # noqa: E731
        us = timeit(hf)
        print("  %-58s %8.1f us  %6.2f TB/s   graph replay %6.1f us" % ("32->1 head: " + what, us, x.numel() * 4 / us / 1e6, timeit_graph(hf)))
    lib.dmb_dev_set_option(24, 0)
