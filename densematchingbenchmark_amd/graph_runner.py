"""Eval-mode forward of a model captured ONCE per input signature in a HIP graph and replayed (hipGraphLaunch).

Why: the reference serves and publishes at batch 1 (dmb/apis/inference.py:191-225 runs ONE pair per call; ResultOfPSMNet.md:15-19
quotes 384x1248 at B = 1).  At batch 4 of 544x960 a step is 46 long launches and the host's launch path hides behind them; at batch
1 of a small image the same 46 launches (plus ~140 of the backbone) are a few microseconds of device work each and the step is
bound by the host issuing them -- ctypes call, argument checks, allocator -- at ~15-25 us per launch.  Every entry point of
libdmb_hip.so only enqueues launches that depend on nothing but their arguments (include/dmb_hip.h, boundary rules), so the whole
forward -- side streams of the backbone's two views and of the first layer included -- records into one graph; a replay costs the
host one call.

``GraphedForward(model)(batch)`` returns the same ``(results, losses)`` as ``model(batch)`` in eval mode, bit for bit (same kernels,
same operands, same order: tests/test_graph_gpu.py).  The returned tensors live in the graph's memory pool and are OVERWRITTEN by
the next call with the same signature: consume (or clone) them before calling again -- the contract of any static-buffer
runner.  ``auto_threshold``: ``wants_graph(batch)`` says whether a batch is small enough for the launch path to matter (the
callers that default to graphs -- apis.inference_stereo, bench.py's latency legs -- ask it)."""
import torch

# Below this many input elements per call (both views together) the host launch path is a measurable share of the step:
# two 3 x 544 x 960 images at batch 1 = 3.1 M elements -> graph; batch 4 of them = 12.5 M -> eager (measured equal there:
# 26.91 against 26.96 ms, profiles/r04 graph_replay_probe).  Feature inputs (32 channels at quarter resolution) count the same way.
AUTO_GRAPH_MAX_ELEMENTS = 2 * 3 * 544 * 960 * 2


def _flatten(obj, prefix=""):
    """[(key path, tensor)] of a batch dict whose values are tensors, tuples / lists of tensors, or anything else (ignored)."""
    out = []
    if torch.is_tensor(obj):
        out.append((prefix, obj))
    elif isinstance(obj, dict):
        for k in sorted(obj):
            out += _flatten(obj[k], "%s/%s" % (prefix, k))
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            out += _flatten(v, "%s/%d" % (prefix, i))
    return out


def _rebuild(obj, mapping, prefix=""):
    if torch.is_tensor(obj):
        return mapping[prefix]
    if isinstance(obj, dict):
        return {k: _rebuild(v, mapping, "%s/%s" % (prefix, k)) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_rebuild(v, mapping, "%s/%d" % (prefix, i)) for i, v in enumerate(obj))
    return obj


def wants_graph(batch, max_elements=AUTO_GRAPH_MAX_ELEMENTS):
    """The default policy: graph replay for CUDA batches of at most ``max_elements`` input elements."""
    ts = [t for _, t in _flatten(batch)]
    return bool(ts) and all(t.is_cuda for t in ts) and sum(t.numel() for t in ts) <= max_elements


class GraphedForward:
    def __init__(self, model, warmup=2, max_graphs=4, track_parameters=True):
        """``max_graphs``: graphs held at once (one per input signature, least recently used evicted).  Each pins a private
        memory pool with every intermediate of a forward -- for a 544x960 pair the three full-resolution cost volumes alone are
        1.2 GB -- until it is evicted or ``reset()`` is called.  ``track_parameters``: compare every parameter's / buffer's (pointer, version) before each replay (~0.15 ms of host
        time for PSMNet's 517 tensors) and re-capture after a change; False = the caller promises frozen weights (or calls
        ``reset()`` after changing them)."""
        self.model = model
        self.warmup = int(warmup)
        self.max_graphs = int(max_graphs)
        self.track_parameters = bool(track_parameters)
        self._tensors = None
        self._graphs = {}      # signature -> (graph, static inputs {path: tensor}, outputs)
        self._order = []
        self._params_key = None

    def _signature(self, flat):
        return tuple((p, tuple(t.shape), t.dtype, t.device) for p, t in flat)

    def _parameters_key(self):
        """Captured graphs hold the PACKED weights of the moment of capture: any in-place change of a parameter or buffer
        (load_state_dict, an optimiser step) invalidates them."""
        if self._tensors is None:      # the Parameter / buffer OBJECTS persist across .to() and load_state_dict(); walking the module
            self._tensors = list(self.model.parameters()) + list(self.model.buffers())     # tree costs 1.6 ms, this list 0.15 ms
        return tuple((t.data_ptr(), t._version) for t in self._tensors)

    def reset(self):
        self._graphs.clear()
        self._order.clear()

    def _capture(self, batch, flat):
        if self.model.training:
            raise RuntimeError("GraphedForward captures the eval-mode forward only")
        static = {p: t.clone() for p, t in flat}
        sbatch = _rebuild(batch, static)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():       # off the capturing stream: weight packing, per-device attributes, side streams
            for _ in range(max(1, self.warmup)):
                self.model(sbatch)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph), torch.no_grad():
            out = self.model(sbatch)
        return graph, static, out

    def __call__(self, batch):
        flat = _flatten(batch)
        if not flat or not all(t.is_cuda for _, t in flat):
            return self.model(batch)      # (raises the library's own error for CPU tensors)
        if self.track_parameters:
            pk = self._parameters_key()
            if pk != self._params_key:
                self.reset()
                self._params_key = pk
        sig = self._signature(flat)
        entry = self._graphs.get(sig)
        if entry is None:
            if len(self._order) >= self.max_graphs:
                self._graphs.pop(self._order.pop(0), None)
            entry = self._capture(batch, flat)
            self._graphs[sig] = entry
            self._order.append(sig)
        elif self._order[-1] != sig:       # least-recently-USED eviction: a hit moves the signature to the young end
            self._order.remove(sig)
            self._order.append(sig)
        graph, static, out = entry
        for p, t in flat:
            if static[p].data_ptr() != t.data_ptr():
                static[p].copy_(t, non_blocking=True)
        graph.replay()
        return out
