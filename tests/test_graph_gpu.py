"""The library's launch contract (include/dmb_hip.h, boundary rules): every entry point only enqueues kernel launches that
depend on nothing but their arguments -- so a whole step can be captured into a HIP graph and replayed, and launches from
several host threads on distinct streams do not interfere.  Round 3's transposed convolution broke both (a host-predicted
counter base as a kernel argument, a lazily allocated counter ring); its counters now live in a caller-provided workspace the
kernel itself resets."""
import os
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib_ws_ints():
    from densematchingbenchmark_amd import _lib
    return _lib.DECONV3D_WORKSPACE_BYTES // 4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(dev, cfg_rel="PSMNet/scene_flow.py"):
    from densematchingbenchmark_amd import synthetic
    from densematchingbenchmark_amd.config import Config
    from densematchingbenchmark_amd.modeling import build_model
    cfg = Config.fromfile(os.path.join(ROOT, "configs", cfg_rel))
    model = build_model(cfg, backbone=None).eval()
    synthetic.init_params_(model, seed=0, classif_gain=10.0)
    return model.to(dev)


def test_psmnet_step_captured_in_a_hip_graph_replays_bit_identically(dev):
    """One PSMNet step at 544x960 / max_disp 192 (two pairs) captured in torch.cuda.CUDAGraph (= hipGraph), replayed three times
    on fresh inputs: disparity maps AND full-resolution costs equal, bit for bit, to the eager evaluation of the same inputs."""
    from densematchingbenchmark_amd import synthetic
    model = _model(dev)
    B = 2
    left, right = synthetic.feature_batch(0, 1, B, 32, 136, 240, dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():      # warm-up off the default stream: weight packing, per-device attributes
        for _ in range(2):
            model(dict(leftFeature=left, rightFeature=right))
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph), torch.no_grad():
        results, _ = model(dict(leftFeature=left, rightFeature=right))
        captured = list(results["disps"]) + list(results["costs"])
    for rep in range(3):
        l2, r2 = synthetic.feature_batch(10 + 2 * rep, 1, B, 32, 136, 240, dev)
        left.copy_(l2)
        right.copy_(r2)
        graph.replay()
        torch.cuda.synchronize()
        got = [t.clone() for t in captured]
        with torch.no_grad():
            eager, _ = model(dict(leftFeature=l2, rightFeature=r2))
        want = list(eager["disps"]) + list(eager["costs"])
        for k, (a, b) in enumerate(zip(got, want)):
            assert torch.equal(a, b), (rep, k, (a - b).abs().max().item())
        del eager, want, got


@pytest.mark.parametrize("cfg_rel,shape", [("AcfNet/kitti_2015_adaptive.py", (96, 312)), ("StereoNet/scene_flow_8x_2stage.py", (48, 156)),
                                           ("GwcNet/scene_flow.py", (136, 240))])
def test_other_configurations_capture_too(dev, cfg_rel, shape):
    """AcfNet with its confidence network at the KITTI shape (learned up-sampling, composed confidence heads, the row-padded
    deepest level), the StereoNet cost path and the GwcNet-style volume: one captured step replayed on fresh inputs equals the
    eager evaluation bit for bit -- no entry point on these paths synchronises, allocates device memory behind the caller's
    back or reads a result back to the host after the first (warm-up) call."""
    from densematchingbenchmark_amd import synthetic
    model = _model(dev, cfg_rel)
    fh, fw = shape
    gwc = cfg_rel.startswith("GwcNet")

    def features(first):
        if gwc:
            lg, rg = synthetic.feature_batch(first, 1, 1, 320, fh, fw, dev)
            lc, rc = synthetic.feature_batch(first + 100000, 1, 1, 12, fh, fw, dev)
            return [lg, lc, rg, rc]
        return list(synthetic.feature_batch(first, 1, 1, 32, fh, fw, dev))

    def run(ts):
        batch = dict(leftFeature=(ts[0], ts[1]), rightFeature=(ts[2], ts[3])) if gwc else dict(leftFeature=ts[0], rightFeature=ts[1])
        res, _ = model(batch)
        return list(res["disps"]) + list(res["costs"]) + list(res.get("confs", []))

    static = features(0)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        for _ in range(2):
            run(static)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph), torch.no_grad():
        captured = run(static)
    for rep in range(2):
        fresh = features(20 + rep)
        for a, b in zip(static, fresh):
            a.copy_(b)
        graph.replay()
        torch.cuda.synchronize()
        got = [t.clone() for t in captured]
        with torch.no_grad():
            want = run(fresh)
        assert len(got) == len(want)
        for k, (a, b) in enumerate(zip(got, want)):
            assert torch.equal(a, b), (cfg_rel, rep, k, (a - b).abs().max().item())


def test_deconv3d_replays_from_a_graph_many_times(dev):
    """The transposed convolution alone, 50 replays: every replay finds the workspace zeroed by the previous one."""
    from densematchingbenchmark_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn((2, 64, 6, 17, 60), generator=g).to(dev)
    w = (torch.randn((64, 32, 3, 3, 3), generator=g) * 0.05).to(dev)
    res = torch.randn((2, 32, 12, 34, 120), generator=g).to(dev)
    wp = ops.pack_deconv3d_weights(w)
    ws = torch.zeros(_lib_ws_ints(), dtype=torch.int32, device=dev)
    want = ops.deconv3d_k3s2(x, wp, 32, None, None, res, True, workspace=None)      # the form without counters
    out = torch.empty_like(want)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ops.deconv3d_k3s2(x, wp, 32, None, None, res, True, workspace=ws, out=out)
    for rep in range(50):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want), rep
        assert int(ws.abs().sum().item()) == 0


def test_deconv3d_from_two_host_threads_on_two_streams(dev):
    """dmb_deconv3d_k3s2_f32 called concurrently from two host threads, each on its own stream (ctypes releases the GIL around
    the call) with its own workspace: 40 launches each, every result equal to the sequential one."""
    from densematchingbenchmark_amd import ops
    g = torch.Generator().manual_seed(6)
    cases = []
    for k in range(2):
        x = torch.randn((2, 64, 6, 34, 60), generator=g).to(dev)
        w = (torch.randn((64, 64, 3, 3, 3), generator=g) * 0.05).to(dev)
        res = torch.randn((2, 64, 12, 68, 120), generator=g).to(dev)
        wp = ops.pack_deconv3d_weights(w)
        cases.append((x, wp, res, ops.deconv3d_k3s2(x, wp, 64, None, None, res, True, workspace=None)))
    torch.cuda.synchronize()
    errors = []

    def worker(k):
        try:
            torch.cuda.set_device(dev)
            x, wp, res, want = cases[k]
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                outs = [ops.deconv3d_k3s2(x, wp, 64, None, None, res, True) for _ in range(40)]
            st.synchronize()
            for i, o in enumerate(outs):
                if not torch.equal(o, want):
                    errors.append((k, i, (o - want).abs().max().item()))
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]
    # the two streams got two workspaces
    assert len({ws.data_ptr() for ws in ops._deconv_ws.values()}) == len(ops._deconv_ws) >= 2


def test_two_graphs_captured_before_either_is_replayed(dev):
    """The host layer's per-stream workspace cache must not hand a graph a workspace whose zero fill is a node of ANOTHER graph
    that has not run yet: capture A, capture B on the same capture stream, replay only B (then A)."""
    from densematchingbenchmark_amd import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn((1, 64, 4, 9, 60), generator=g).to(dev)
    w = (torch.randn((64, 64, 3, 3, 3), generator=g) * 0.05).to(dev)
    wp = ops.pack_deconv3d_weights(w)
    want = ops.deconv3d_k3s2(x, wp, 64, None, None, None, True, workspace=None)
    outs = [torch.zeros_like(want) for _ in range(2)]
    graphs = []
    torch.cuda.synchronize()
    for k in range(2):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            ops.deconv3d_k3s2(x, wp, 64, None, None, None, True, out=outs[k])      # workspace="auto"
        graphs.append(gr)
    for k in (1, 0, 1):
        outs[k].zero_()
        graphs[k].replay()
        torch.cuda.synchronize()
        assert torch.equal(outs[k], want), k


def test_backbone_with_its_two_view_streams_captures(dev):
    """The PSMNet backbone runs the two views on two streams (ops.two_view_forward: fork / join by events) -- a capture on the
    caller's stream has to pick the side stream up through those events and replay both chains."""
    from densematchingbenchmark_amd import ops, synthetic
    from densematchingbenchmark_amd.modeling.stereo.backbones import PSMNetBackbone
    bb = PSMNetBackbone(3, True).eval()
    synthetic.init_params_(bb, seed=8, classif_gain=1.0)
    bb = bb.to(dev)
    g = torch.Generator().manual_seed(3)
    l, r = (torch.randn((1, 3, 256, 512), generator=g).to(dev) for _ in range(2))
    assert ops.view_streams()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        bb(l, r)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph), torch.no_grad():
        fl, fr = bb(l, r)
    for rep in range(2):
        l2, r2 = (torch.randn((1, 3, 256, 512), generator=g).to(dev) for _ in range(2))
        l.copy_(l2)
        r.copy_(r2)
        graph.replay()
        torch.cuda.synchronize()
        got = (fl.clone(), fr.clone())
        with torch.no_grad():
            want = bb(l2, r2)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), rep
