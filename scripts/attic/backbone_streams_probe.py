"""Development aid (round 4): the PSMNet backbone's two views as ONE batch of 2B images (the default) against two independent
chains of B images on two HIP streams.  A 64-channel layer of 8 images is 1360 tiles on 512 persistent slots = 2.66 rounds (88 %
fill, and 5.3 tiles per CU however they are dealt); two chains in flight fill each other's partial rounds."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from densematchingbenchmark_amd import ops, synthetic
from densematchingbenchmark_amd.modeling.stereo.backbones import PSMNetBackbone

dev = torch.device("cuda:0")
B, H, W = int(os.environ.get("KB_B", "4")), 544, 960
bb = PSMNetBackbone(3, True).eval()
synthetic.init_params_(bb, seed=8, classif_gain=1.0)
bb = bb.to(dev)
g = torch.Generator().manual_seed(77)
l, r = (torch.randn((B, 3, H, W), generator=g).to(dev) for _ in range(2))
side = ops.side_stream(dev)


def one_batch():
    return bb(l, r)


def two_streams():
    main = torch.cuda.current_stream(dev)
    fork = main.record_event()
    with torch.cuda.stream(side):
        side.wait_event(fork)
        fr = bb._forward(r)
        fr.record_stream(main)
        done = side.record_event()
    fl = bb._forward(l)
    main.wait_event(done)
    return fl, fr


def run(fn, n):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


streams = [torch.cuda.Stream(dev) for _ in range(8)]
imgs = torch.cat((l, r), 0)


def chains(n):
    """the 2B images as n independent chains of 2B / n images, chain k on stream k (chain 0 on the caller's)"""
    def fn():
        main = torch.cuda.current_stream(dev)
        fork = main.record_event()
        per = 2 * B // n
        outs, evs = [None] * n, []
        for k in range(1, n):
            with torch.cuda.stream(streams[k]):
                streams[k].wait_event(fork)
                outs[k] = bb._forward(imgs[k * per:(k + 1) * per])
                outs[k].record_stream(main)
                evs.append(streams[k].record_event())
        outs[0] = bb._forward(imgs[:per])
        for e in evs:
            main.wait_event(e)
        return outs
    return fn


with torch.no_grad():
    a, b = one_batch(), two_streams()
    torch.cuda.synchronize()
    print("identical:", all(torch.equal(x, y) for x, y in zip(a, b)))
    cases = [("one batch of %d images" % (2 * B), one_batch), ("two streams of %d" % B, two_streams)]
    for n in (4, 8):
        if 2 * B % n == 0:
            cn = torch.cat(chains(n)(), 0)
            torch.cuda.synchronize()
            print("%d chains identical:" % n, torch.equal(cn, torch.cat(a, 0)))
            cases.append(("%d chains" % n, chains(n)))
    run(one_batch, 3)
    acc = {name: [] for name, _ in cases}
    for rep in range(4):
        for name, fn in cases:
            run(fn, 2)
            acc[name].append(run(fn, 8))
for name, ts in acc.items():
    print("%-26s %s  -> min %.3f ms" % (name, " ".join("%.3f" % t for t in ts), min(ts)))
