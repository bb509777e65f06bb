"""Functional layer over the C ABI (include/dmb_hip.h): one Python function per entry point.

Only shape bookkeeping and output allocation happen here (PyTorch as the device allocator); every number is
produced by a HIP kernel of libdmb_hip.so.  All functions raise ``DmbLibraryError`` on CPU tensors -- the
product path has no fallback.
"""
import ctypes

import torch

from . import _lib
from ._lib import check, dev_ptr, host_floats, host_ints, stream_ptr


def disp_index_list(max_disp, start_disp=0, dilation=1):
    """Integer disparity indices exactly as the reference derives them:
    ``int(torch.linspace(start, start + max_disp - 1, n)[k])`` with ``n = (max_disp + dilation - 1) // dilation``
    (cost_processors/utils/cat_fms.py:26-35) -- truncation toward zero of an FP32 linspace."""
    end_disp = start_disp + max_disp - 1
    n = (max_disp + dilation - 1) // dilation
    return [int(v) for v in torch.linspace(start_disp, end_disp, n)]


def disp_sample_values(max_disp, start_disp=0, dilation=1):
    """FP32 disparity sample values of the predictors (disp_predictors/faster_soft_argmin.py:33-44)."""
    end_disp = start_disp + max_disp - 1
    n = (max_disp + dilation - 1) // dilation
    return [float(v) for v in torch.linspace(start_disp, end_disp, n)]


class KernelTimer:
    """HIP-event timing of selected launches on the stream they are enqueued on (torch's current stream is the
    stream handed to the C ABI).  Used by bench.py to measure the dominant kernel's launch duration live inside
    the timed region; disabled (None) otherwise."""

    def __init__(self, tags, every=1):
        self.tags = set(tags)
        self.events = {t: [] for t in self.tags}
        self._open = None
        # an event record costs the stream about 5 us on either side of the launch it brackets: time the launches of every
        # ``every``-th step only (``begin_step`` counts them), so that the observer takes 1 / every of that out of the run
        self.every, self._step, self._on = max(1, int(every)), -1, True

    def begin_step(self):
        self._step += 1
        self._on = self._step % self.every == 0

    def start(self, tag):
        if self._on and tag in self.tags:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            self._open = (tag, e0)

    def stop(self, tag):
        if self._open is not None and self._open[0] == tag:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.events[tag].append((self._open[1], e1))
            self._open = None

    def reset(self):
        self.events = {t: [] for t in self.tags}

    def mean_ms(self, tag):
        ev = self.events[tag]
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev) if ev else float("nan")

    def count(self, tag):
        return len(self.events[tag])


_kernel_timer = None


def set_kernel_timer(timer):
    global _kernel_timer
    _kernel_timer = timer


def _f32c(t, name):
    if t.dtype != torch.float32:
        raise _lib.DmbLibraryError("%s must be float32, got %s" % (name, t.dtype))
    return t.contiguous()


def _same_shape(a, b, what):
    """Operands a kernel indexes with one set of sizes must HAVE those sizes: a mismatch would read out of bounds on the device."""
    if tuple(a.shape) != tuple(b.shape):
        raise _lib.DmbLibraryError("%s: shapes differ, %s vs %s" % (what, tuple(a.shape), tuple(b.shape)))


def _feature_pair(left, right, what):
    left, right = _f32c(left, "reference_fm"), _f32c(right, "target_fm")
    if left.dim() != 4:
        raise _lib.DmbLibraryError("%s: features must be [B, C, H, W], got %s" % (what, tuple(left.shape)))
    _same_shape(left, right, what)
    return left, right


def _affine_ok(scale, shift, Co, what):
    for t, n in ((scale, "scale"), (shift, "shift")):
        if t is not None and t.numel() != Co:
            raise _lib.DmbLibraryError("%s: %s has %d elements for %d output channels" % (what, n, t.numel(), Co))


# ---------------------------------------------------------------------------------------------- volumes
def cat_fms(left, right, disp_idx):
    lib = _lib.load()
    left, right = _feature_pair(left, right, "cat_fms")
    B, C, H, W = left.shape
    D = len(disp_idx)
    out = torch.empty((B, 2 * C, D, H, W), dtype=torch.float32, device=left.device)
    check(lib.dmb_cat_fms_f32(dev_ptr(left), dev_ptr(right), dev_ptr(out), B, C, H, W, D, host_ints(disp_idx),
                              stream_ptr(left.device)), "dmb_cat_fms_f32")
    return out


def dif_fms(left, right, disp_idx):
    lib = _lib.load()
    left, right = _feature_pair(left, right, "dif_fms")
    B, C, H, W = left.shape
    D = len(disp_idx)
    out = torch.empty((B, C, D, H, W), dtype=torch.float32, device=left.device)
    check(lib.dmb_dif_fms_f32(dev_ptr(left), dev_ptr(right), dev_ptr(out), B, C, H, W, D, host_ints(disp_idx),
                              stream_ptr(left.device)), "dmb_dif_fms_f32")
    return out


def fast_disp_samples(max_disp, start_disp=0, dilation=1):
    """cat_fms.py:55-63: the samples the sample-based builders generate themselves -- linspace(start, end, D), FP32, not
    truncated to integers (max_disp 192 with dilation 2 steps by 191/95)."""
    D = (max_disp + dilation - 1) // dilation
    return torch.linspace(start_disp, start_disp + max_disp - 1, D).float()


def _fast_samples(left, disp_sample):
    B, C, H, W = left.shape
    ds = _f32c(disp_sample.to(left.device), "disp_sample")
    if ds.dim() == 1:
        return ds, ds.numel(), 0
    if ds.dim() != 4 or ds.shape[0] != B or tuple(ds.shape[2:]) != (H, W):
        raise _lib.DmbLibraryError("disp_sample must be [D] or [B, D, H, W] matching the features, got %s" % (tuple(ds.shape),))
    return ds, ds.shape[1], 1


def fast_cat_fms(left, right, disp_sample):
    """cat_fms.py:51-82 on csrc/warp_volume.hip: [B, C, H, W] x 2, samples [D] or [B, D, H, W] -> [B, 2C, D, H, W]."""
    lib = _lib.load()
    left, right = _feature_pair(left, right, "fast_cat_fms")
    B, C, H, W = left.shape
    ds, D, per_pixel = _fast_samples(left, disp_sample)
    out = torch.empty((B, 2 * C, D, H, W), dtype=torch.float32, device=left.device)
    check(lib.dmb_fast_cat_fms_f32(dev_ptr(left), dev_ptr(right), dev_ptr(ds), dev_ptr(out), B, C, D, H, W, per_pixel,
                                   stream_ptr(left.device)), "dmb_fast_cat_fms_f32")
    return out


def fast_dif_fms(left, right, disp_sample, normalize=False, p=1.0):
    """dif_fms.py:49-86: -> [B, C, D, H, W], or [B, D, H, W] (p-norm over the channels) with ``normalize``."""
    lib = _lib.load()
    left, right = _feature_pair(left, right, "fast_dif_fms")
    B, C, H, W = left.shape
    ds, D, per_pixel = _fast_samples(left, disp_sample)
    shape = (B, D, H, W) if normalize else (B, C, D, H, W)
    out = torch.empty(shape, dtype=torch.float32, device=left.device)
    check(lib.dmb_fast_dif_fms_f32(dev_ptr(left), dev_ptr(right), dev_ptr(ds), dev_ptr(out), B, C, D, H, W, per_pixel,
                                   1 if normalize else 0, float(p), stream_ptr(left.device)), "dmb_fast_dif_fms_f32")
    return out


def fast_fms_bwd(left, right, disp_sample, dvol, dif=False, norm_out=None, p=1.0, wrt_samples=False):
    """Backward of fast_cat_fms / fast_dif_fms: (d left, d right, d disp_sample); the first two [B, C, H, W], the third
    [B, D, H, W] with ``wrt_samples`` (per-pixel samples only) and None otherwise.  ``norm_out`` = the forward's output of
    fast_dif_fms(normalize=True, p) selects the normalised form (grad_output [B, D, H, W])."""
    lib = _lib.load()
    left, right = _feature_pair(left, right, "fast_fms_bwd")
    ds, D, per_pixel = _fast_samples(left, disp_sample)
    dvol = _f32c(dvol, "grad_output")
    B, C, H, W = left.shape
    mode = 2 if norm_out is not None else (1 if dif else 0)
    want = (B, D, H, W) if mode == 2 else (B, C if dif else 2 * C, D, H, W)
    if tuple(dvol.shape) != want:
        raise _lib.DmbLibraryError("fast_fms_bwd: grad_output is %s, expected %s" % (tuple(dvol.shape), want))
    if mode == 2:
        norm_out = _f32c(norm_out, "norm_out")
        if tuple(norm_out.shape) != want:
            raise _lib.DmbLibraryError("fast_fms_bwd: norm_out is %s, expected %s" % (tuple(norm_out.shape), want))
    if wrt_samples and not per_pixel:
        raise _lib.DmbLibraryError("fast_fms_bwd: a gradient for the samples needs per-pixel samples [B, D, H, W]")
    dl, dr = torch.empty_like(left), torch.empty_like(right)
    dsamp = torch.empty((B, D, H, W), dtype=torch.float32, device=left.device) if wrt_samples else None
    part = torch.empty((B, C, H, 2, W), dtype=torch.float32, device=left.device)
    check(lib.dmb_fast_fms_bwd_f32(dev_ptr(left), dev_ptr(right), dev_ptr(ds), dev_ptr(dvol), dev_ptr(norm_out, "norm_out", True), dev_ptr(dl),
                                   dev_ptr(dr), dev_ptr(dsamp, "d_samples", True), dev_ptr(part), B, C, D, H, W, per_pixel, mode, float(p),
                                   stream_ptr(left.device)), "dmb_fast_fms_bwd_f32")
    return dl, dr, dsamp


def gwc_fms(left, right, disp_idx, num_groups, out=None, out_ch_offset=0):
    lib = _lib.load()
    left, right = _feature_pair(left, right, "gwc_fms")
    B, C, H, W = left.shape
    D = len(disp_idx)
    if out is None:
        out = torch.empty((B, num_groups, D, H, W), dtype=torch.float32, device=left.device)
    check(lib.dmb_gwc_fms_f32(dev_ptr(left), dev_ptr(right), dev_ptr(out), B, C, num_groups, H, W, D,
                              host_ints(disp_idx), out.shape[1], out_ch_offset, stream_ptr(left.device)),
          "dmb_gwc_fms_f32")
    return out


def correlation1d(left, right, max_disp, negative_slope=0.1):
    """correlation1d_cost.py:7-27: [B, C, H, W] x 2 -> [B, max_disp, H, W], channel j = disparity max_disp - 1 - j."""
    lib = _lib.load()
    left, right = _feature_pair(left, right, "correlation1d")
    B, C, H, W = left.shape
    out = torch.empty((B, max_disp, H, W), dtype=torch.float32, device=left.device)
    check(lib.dmb_correlation1d_f32(dev_ptr(left), dev_ptr(right), dev_ptr(out), B, C, H, W, max_disp, negative_slope,
                                    stream_ptr(left.device)), "dmb_correlation1d_f32")
    return out


def cat_fms_into(left, right, disp_idx, out, out_ch_offset):
    lib = _lib.load()
    left, right = _feature_pair(left, right, "cat_fms_into")
    B, C, H, W = left.shape
    check(lib.dmb_cat_fms_into_f32(dev_ptr(left), dev_ptr(right), dev_ptr(out), B, C, H, W, len(disp_idx),
                                   host_ints(disp_idx), out.shape[1], out_ch_offset, stream_ptr(left.device)),
          "dmb_cat_fms_into_f32")
    return out


# ---------------------------------------------------------------------------------------------- convs
def pack_conv3d_weights(w):
    """nn.Conv3d weight [Co, Ci, 3, 3, 3] -> MFMA A-fragment stream."""
    lib = _lib.load()
    w = _f32c(w, "weight")
    Co, Ci = w.shape[0], w.shape[1]
    wp = torch.empty((lib.dmb_conv3d_packed_floats(Co, Ci),), dtype=torch.float32, device=w.device)
    check(lib.dmb_conv3d_pack_weights_f32(dev_ptr(w), dev_ptr(wp), Co, Ci, stream_ptr(w.device)),
          "dmb_conv3d_pack_weights_f32")
    return wp


def pack_deconv3d_weights(w):
    """nn.ConvTranspose3d weight [Ci, Co, 3, 3, 3] -> MFMA A-fragment stream."""
    lib = _lib.load()
    w = _f32c(w, "weight")
    Ci, Co = w.shape[0], w.shape[1]
    wp = torch.empty((lib.dmb_deconv3d_packed_floats(Ci, Co),), dtype=torch.float32, device=w.device)
    check(lib.dmb_deconv3d_pack_weights_f32(dev_ptr(w), dev_ptr(wp), Ci, Co, stream_ptr(w.device)),
          "dmb_deconv3d_pack_weights_f32")
    return wp


def _relu_mode(relu):
    """False/0: none; True/1: ReLU after the residual add; 'pre'/2: ReLU before it (GC-Net's skip connections)."""
    if relu == "pre":
        return 2
    return int(relu) if relu in (0, 1, 2) else int(bool(relu))


# ---------------------------------------------------------------------------------------------- small-launch policy
# A convolution launch that would leave most of the chip idle (one small pair per call) takes a split-K kernel form: the input
# channels of a voxel split over the waves of a workgroup, partial sums added in a fixed order (include/dmb_hip.h,
# DMB_CONV_SINGLE_CHAIN; csrc/conv3d_sk.hip).  Reproducible run to run, but the last bits then depend on the launch's SIZE: the
# same pair at batch 1 and inside a batch of 4 may differ by an FP32 rounding.  ``set_split_k(False)`` keeps every launch on the
# single-chain kernels: results bit-identical across batch sizes, small launches 2-3x slower.
_split_k = True


def set_split_k(flag):
    """True / "auto" (default): small launches may take the split-K forms; False: single-chain kernels only (batch-invariant bits)."""
    global _split_k
    _split_k = bool(flag)


def split_k():
    return _split_k


def _conv_flags():
    return 0 if _split_k else _lib.CONV_SINGLE_CHAIN


def _out_tensor(out, shape, device, what):
    """The caller's pre-allocated output (as the reference's native op takes it, ops/spn/functions/gaterecurrent2dnoind.py:8-39)
    or a fresh one."""
    if out is None:
        return torch.empty(shape, dtype=torch.float32, device=device)
    if tuple(out.shape) != tuple(shape) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != device:
        raise _lib.DmbLibraryError("%s: out must be a contiguous float32 tensor of shape %s on %s" % (what, tuple(shape), device))
    return out


def conv3d_k3(x, wpack, Co, scale=None, shift=None, residual=None, stride=1, relu=False, out=None):
    sh = _lib.shim()
    if sh is not None:      # the torch-extension shim: the same checks and the same C-ABI call, without the interpreter
        if _kernel_timer is not None:
            _kernel_timer.start("conv3d_k3_s%d_%dto%d" % (stride, x.shape[1], Co))
        y = sh.conv3d_k3(x if x.is_contiguous() else x.contiguous(), wpack, Co, scale, shift, residual, stride,
                         _relu_mode(relu) | _conv_flags(), out)
        if _kernel_timer is not None:
            _kernel_timer.stop("conv3d_k3_s%d_%dto%d" % (stride, x.shape[1], Co))
        return y
    lib = _lib.load()
    x = _f32c(x, "x")
    B, Ci, D, H, W = x.shape
    Do, Ho, Wo = (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1
    y = _out_tensor(out, (B, Co, Do, Ho, Wo), x.device, "conv3d_k3")
    if residual is not None and tuple(residual.shape) != tuple(y.shape):
        raise _lib.DmbLibraryError("residual shape %s != output shape %s" % (tuple(residual.shape), tuple(y.shape)))
    _affine_ok(scale, shift, Co, "conv3d_k3")
    if wpack.numel() != lib.dmb_conv3d_packed_floats(Co, Ci):
        raise _lib.DmbLibraryError("conv3d_k3: packed weights hold %d floats, %d -> %d channels need %d"
                                   % (wpack.numel(), Ci, Co, lib.dmb_conv3d_packed_floats(Co, Ci)))
    tag = "conv3d_k3_s%d_%dto%d" % (stride, Ci, Co)
    if _kernel_timer is not None:
        _kernel_timer.start(tag)
    check(lib.dmb_conv3d_k3_f32(dev_ptr(x), dev_ptr(wpack), dev_ptr(scale, allow_none=True),
                                dev_ptr(shift, allow_none=True), dev_ptr(residual, allow_none=True), dev_ptr(y),
                                B, Ci, Co, D, H, W, stride, _relu_mode(relu) | _conv_flags(), stream_ptr(x.device)), "dmb_conv3d_k3_f32")
    if _kernel_timer is not None:
        _kernel_timer.stop(tag)
    return y


# ---------------------------------------------------------------------------------------------- first layer on a cat volume
# The aggregators' first convolution (2C -> Co, k3 s1 p1) applied to a concatenation volume with unit disparity step
# collapses into 2-D maps (csrc/catconv.hip): the volume is never written, 2/3 of the layer's multiplications vanish.
# The module layer (FusedConv3d.forward_cat) decides when this applies; "off" forces the materialised volume everywhere.
_cat_fusion = True


def set_cat_fusion(flag):
    global _cat_fusion
    _cat_fusion = bool(flag)


def cat_fusion():
    return _cat_fusion


# Branch overlap (eval): the classifier branch of hourglass k (two full-resolution convolutions, the up-sampling and the
# regression) depends on that hourglass's output only, and so does hourglass k + 1 -- whose quarter-resolution layers cannot fill
# the chip.  With the switch on, the aggregators issue the branch on a second HIP stream (fork / join with events):
# same kernels, same operands, identical results; 27.28 -> 27.03 ms per BASELINE step (scripts/overlap_probe.py).  The
# opposite arrangement -- the dependent chain on a high-priority stream, the branches on the caller's -- gains half as much.
# Default "auto" (round 5): on for launches of at most AUTO_OVERLAP_MAX_VOXELS quarter-resolution voxels -- ONE pair of up to
# 544x960 / max_disp 192, the serving regime (dmb/apis/inference.py: one pair per call), where no layer fills the chip and the
# branch's kernels run in the gaps of the dependent chain: 1.72 -> 1.62 ms at 256x512 / D 64, 7.48 -> 7.30 ms at 384x1248, 7.71 ->
# 7.65 ms at 544x960 (profiles/r05_cosched_probe.log) -- and off for anything larger.  Under concurrency the per-kernel durations
# that the roofline accounting rests on (HIP events in bench.py, rocprofv3 --stats) no longer describe one kernel alone, so bench.py's
# timed loop switches it OFF explicitly and reports the overlapped batch-4 step as a secondary leg.
_branch_overlap = "auto"
AUTO_OVERLAP_MAX_VOXELS = 48 * 136 * 240
_side_streams = {}


def set_branch_overlap(flag):
    """True / False, or "auto" (default): on for one small pair, off for batches (see above)."""
    global _branch_overlap
    _branch_overlap = "auto" if flag == "auto" else bool(flag)


def branch_overlap(raw_cost=None):
    """The switch's value for a launch on the volume ``raw_cost`` ([B, C, D, H, W] tensor or description)."""
    if _branch_overlap != "auto":
        return _branch_overlap
    if raw_cost is None:
        return False
    B, _, D, H, W = raw_cost.shape
    return B * D * H * W <= AUTO_OVERLAP_MAX_VOXELS


def side_stream(device, which=0):
    """Extra HIP streams per device (``which`` = 0, 1, ...) for independent launches that run next to the caller's stream."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), which)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device)
    return _side_streams[key]


# The two views of a stereo pair go through a shared-weight backbone independently.  As ONE batch of 2B images a layer's tiles
# rarely divide over the persistent grid (8 images of 136 x 240 at 64 channels: 1360 tiles on 512 slots = 2.66 rounds, 5.3 tiles
# per CU however they are dealt); as two chains of B images on two streams the partial rounds of one chain are filled by the
# other's.  Measured on the PSMNet backbone (scripts/attic/backbone_streams_probe.py): 16.15 -> 14.67 ms at B = 4, 5.60 -> 5.14 at
# B = 1, 30.4 -> 28.6 at B = 8; four or eight chains are slower than one batch.  Same launches per image: identical results.
_view_streams = True


def set_view_streams(flag):
    global _view_streams
    _view_streams = bool(flag)


def view_streams():
    return _view_streams


# Folded / packed parameter caches of the modules are keyed by (pointer, ``_version``) of their parameters -- and by this epoch.
# Most in-place updates move ``_version`` (optimizer.step() of the for-loop / multi-tensor optimizers, load_state_dict, copy_), but
# torch's FUSED optimizers (``Adam(fused=True)``) do not.  Every train() <-> eval() switch of a fused unit advances the epoch, so
# an eval-mode forward after training re-folds whatever the versions say; code that rewrites parameters of an eval-mode model
# behind torch's back calls ``bump_param_epoch()`` itself.
_param_epoch = [0]


def param_epoch():
    return _param_epoch[0]


def bump_param_epoch():
    _param_epoch[0] += 1


def warm_packed_parameters(module):
    """Fill (on the CURRENT stream) every lazily packed / folded parameter cache below ``module``: each fused unit keeps its packed
    weights and folded BatchNorm affine keyed by the parameters' versions (``_prepacked()``) and refills them inside its first
    forward after a change -- pack kernels and torch ops on whatever stream that forward runs on.
    Cheap when nothing changed: the walk over the module tree (1.6 ms for PSMNet's backbone) only happens when the (pointer, version)
    key of the module's parameters and buffers differs from the one of the last warm-up (0.1 ms to compute)."""
    tensors = module.__dict__.get("_dmb_warm_tensors")
    if tensors is None:
        tensors = list(module.parameters()) + list(module.buffers())     # the objects persist across .to() and load_state_dict()
        module.__dict__["_dmb_warm_tensors"] = tensors
    key = (module.training, _param_epoch[0]) + tuple((t.data_ptr(), t._version) for t in tensors)
    if module.__dict__.get("_dmb_warm_key") == key:
        return
    for m in module.modules():
        pre = getattr(m, "_prepacked", None)
        if pre is not None:
            pre()
    module.__dict__["_dmb_warm_key"] = key


def two_view_forward(fn, left, right, module=None):
    """``(fn(left), fn(right))`` for a per-image function (an eval-mode backbone): the right view on a side stream of the device,
    forked from and joined back into the caller's stream with events; ``ops.set_view_streams(False)``: one batch of both views.
    ``module``: the nn.Module ``fn`` evaluates.  Its packed-weight caches are filled on the CALLER's stream before the fork:
    filled lazily inside ``fn(right)`` they would be written by pack kernels on the side stream while ``fn(left)`` -- which finds
    the host-side cache entry already present -- reads them on the caller's stream with nothing ordering the two (cold caches:
    the first forward, after load_state_dict, after a training step)."""
    if not (_view_streams and left.is_cuda):
        B = left.shape[0]
        f = fn(torch.cat((left, right), 0))
        return f[:B], f[B:]
    if module is not None:
        warm_packed_parameters(module)
    main = torch.cuda.current_stream(left.device)
    side = side_stream(left.device, 2)
    fork = main.record_event()
    with torch.cuda.stream(side):
        side.wait_event(fork)
        fr = fn(right)
        fr.record_stream(main)          # allocated on the side stream, consumed on the caller's
        right.record_stream(side)       # ... and the caller's input is read there
        done = side.record_event()
    fl = fn(left)
    main.wait_event(done)
    return fl, fr


_first_layer_mode = "merged"


def set_first_layer_mode(mode):
    """How catconv_first issues its five 2-D convolutions: "merged" (one multi-job launch, default), "streams" (the three border
    maps on two side streams) or "serial" (five launches on one stream).  Same kernels' arithmetic, identical results."""
    global _first_layer_mode
    if mode not in ("merged", "streams", "serial"):
        raise ValueError("first-layer mode must be 'merged', 'streams' or 'serial'")
    _first_layer_mode = mode


def set_first_layer_streams(flag):
    """(round-3 interim switch, kept for scripts) True -> "streams", False -> "serial"."""
    set_first_layer_mode("streams" if flag else "serial")


def copy_window(src, Wd, xs):
    """dst[..., j] = src[..., j + xs] (zero outside [0, W)), j in [0, Wd): zero-filled column window of a [.., W] tensor."""
    sh = _lib.shim()
    if sh is not None:
        return sh.copy_window(src if src.is_contiguous() else src.contiguous(), int(Wd), int(xs))
    lib = _lib.load()
    src = _f32c(src, "src")
    W = src.shape[-1]
    dst = torch.empty(tuple(src.shape[:-1]) + (Wd,), dtype=torch.float32, device=src.device)
    rows = src.numel() // W
    check(lib.dmb_copy_window_f32(dev_ptr(src), dev_ptr(dst), rows, W, Wd, xs, stream_ptr(src.device)), "dmb_copy_window_f32")
    return dst


def zero_columns_(t, x0):
    """t[..., x0:] = 0 in place: the padding columns of a row-padded tensor (see deconv3d_k3s2, ``out_width``)."""
    lib = _lib.load()
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise _lib.DmbLibraryError("zero_columns_: contiguous float32 tensor expected")
    pitch = t.shape[-1]
    check(lib.dmb_zero_columns_f32(dev_ptr(t), t.numel() // pitch, pitch, int(x0), stream_ptr(t.device)), "dmb_zero_columns_f32")
    return t


def padded_rows_applicable(x, Co):
    """Whether a [B, Ci, D, H, W] tensor whose rows are NOT a 16-byte multiple (W % 4 != 0) can go through a stride-1 unit and a
    transposed unit with its rows padded to the next multiple of 4 (zero padding columns): the transposed kernel's padded-row
    form needs an even W with 2 W % 4 == 0 ... i.e. W % 4 == 2, Co in (32, 64), whole 16-channel chunks."""
    W, Ci = x.shape[-1], x.shape[1]
    if not (x.is_cuda and W % 4 == 2 and Co in (32, 64) and Ci % 16 == 0 and Ci >= 32):
        return False
    # ... and what csrc/deconv3d_zy.hip::deconv3d_zy_try checks besides (it is the only form that takes a row-padded input: if it
    # declined, dmb_deconv3d_k3s2_f32 would fail with DMB_EUNSUPPORTED where the unpadded path has a fallback): 16 input channels of
    # one batch item and one batch item of the output below 2 GiB; 16-byte alignment holds for every tensor the allocator hands out
    D, H = x.shape[2], x.shape[3]
    Wp = (W + 3) // 4 * 4
    return 16 * D * H * Wp * 4 < 2 ** 31 - 1 and Co * 8 * D * H * Wp * 4 < 2 ** 31 - 1 and x.data_ptr() % 16 == 0


CATCONV_CH = 128   # channel count of the per-dz map tensors: 3 * Co used (Co <= 32), the rest are zero-weight rows


def catconv_applicable(L, R, disp_idx, Co):
    """Shapes the 2-D form covers: unit disparity step from 0 (d_k = k), at least 2 planes, Co <= 32, 16-byte rows,
    the image wider than the near-diagonal band."""
    if not _cat_fusion or L.dim() != 4 or L.shape != R.shape:
        return False
    D, W = len(disp_idx), L.shape[3]
    return (list(disp_idx) == list(range(D)) and D >= 4 and D % 4 == 0 and W % 4 == 0 and W >= D + 8 and Co <= 32
            and L.dtype == torch.float32 and L.is_cuda)


def catconv_pack(w, kind="cat"):
    """nn.Conv3d weight -> the five conv2d weight packs of the 2-D form (csrc/catconv.hip): per-dz slices of the left half
    (all dx taps; dx >= 1; dx >= 2) and of the right half (all taps; without dx = 2), stacked along the output-channel axis
    as dz * Co + co and zero-padded to CATCONV_CH rows.  ``kind="cat"``: w is [Co, 2C, 3, 3, 3], left half = channels
    [0, C), right half = [C, 2C).  ``kind="dif"`` (difference volume, L - R shifted: dif_fms.py:7-46): w is [Co, C, 3, 3, 3]
    and convolution is linear, so the left half takes w and the right half -w (a sign flip is exact).
    One launch (dmb_catconv_pack_weights_f32); ``catconv_pack_torch`` is the same thing spelled out with tensor slicing."""
    lib = _lib.load()
    w = _f32c(w.detach(), "weight")
    Co = w.shape[0]
    C = w.shape[1] if kind == "dif" else w.shape[1] // 2
    n = lib.dmb_conv2d_packed_floats(CATCONV_CH, C, 3)
    buf = torch.empty((5, n), dtype=torch.float32, device=w.device)
    check(lib.dmb_catconv_pack_weights_f32(dev_ptr(w), dev_ptr(buf), Co, C, CATCONV_CH, 1 if kind == "dif" else 0, stream_ptr(w.device)),
          "dmb_catconv_pack_weights_f32")
    return {"A": buf[0], "B1": buf[1], "B2": buf[2], "HC": buf[3], "HD": buf[4], "Co": Co, "Cin": C}


def catconv_pack_torch(w, kind="cat"):
    """catconv_pack as tensor algebra + five dmb_conv2d_pack_weights_f32 launches (rounds 2-5; kept as the definition the
    one-launch kernel is tested against)."""
    Co = w.shape[0]
    w = w.detach().float()
    if kind == "dif":
        wl, wr = w, -w
    else:
        C = w.shape[1] // 2
        wl, wr = w[:, :C], w[:, C:]
    C = wl.shape[1]

    def stack(half, dx_from, dx_to):
        k = torch.zeros((CATCONV_CH, C, 3, 3), dtype=torch.float32, device=w.device)
        for dz in range(3):
            k[dz * Co:(dz + 1) * Co, :, :, dx_from:dx_to] = half[:, :, dz, :, dx_from:dx_to]
        return pack_conv2d_weights(k)

    return {"A": stack(wl, 0, 3), "B1": stack(wl, 1, 3), "B2": stack(wl, 2, 3), "HC": stack(wr, 0, 3), "HD": stack(wr, 0, 2),
            "Co": Co, "Cin": C}


def catconv_first(L, R, D, packs, scale=None, shift=None, relu=False):
    """relu?(scale * conv3d(V, w) + shift) without the volume V = cat_fms(L, R, d_k = k) (or dif_fms: the packs decide):
    [B, C, H, W] x 2 -> [B, Co, D, H, W]."""
    lib = _lib.load()
    L, R = _feature_pair(L, R, "catconv_first")
    B, C, H, W = L.shape
    if packs.get("Cin") is not None and packs["Cin"] != C:
        raise _lib.DmbLibraryError("catconv_first: weights were packed for %d feature channels, the maps have %d" % (packs["Cin"], C))
    Co, CA = packs["Co"], CATCONV_CH
    Wc = D + 4
    dev = L.device
    # The five 2-D convolutions are independent and three of them are small (the 52-column border maps: 272 tiles each
    # for 512 workgroup slots): one after the other they take six rounds of the chip where their arithmetic fills less than
    # four.  ``set_first_layer_mode``: "merged" (default) = ONE multi-job launch walks all their tiles; "streams" = the border
    # maps on two side streams next to the two big ones; "serial" = five launches on one stream.  Same arithmetic every way.
    Lc, Rw, Rl = copy_window(L, Wc, 0), copy_window(R, W + 4, -4), copy_window(R, Wc, W - Wc)
    if _first_layer_mode == "merged":
        FA = torch.empty((B, CA, H, W), dtype=torch.float32, device=dev)              # F_{dz, 0}
        HC = torch.empty((B, CA, H, W + 4), dtype=torch.float32, device=dev)          # H_dz at n = j - 4
        FB = torch.empty((B, 2 * CA, H, Wc), dtype=torch.float32, device=dev)         # F_{dz, 1} | F_{dz, 2}
        HD = torch.empty((B, CA, H, Wc), dtype=torch.float32, device=dev)             # border variant
        conv2d_k3_multi([(Rw, packs["HC"], HC, 0), (L, packs["A"], FA, 0), (Lc, packs["B1"], FB, 0), (Lc, packs["B2"], FB, CA),
                         (Rl, packs["HD"], HD, 0)], CA)
    elif _first_layer_mode == "streams":
        main = torch.cuda.current_stream(dev)
        s1, s2 = side_stream(dev), side_stream(dev, 1)
        fork = main.record_event()
        with torch.cuda.stream(s1):
            s1.wait_event(fork)
            FB = torch.empty((B, 2 * CA, H, Wc), dtype=torch.float32, device=dev)
            conv2d(Lc, packs["B1"], CA, 3, out=FB, out_ch_offset=0)
            conv2d(Lc, packs["B2"], CA, 3, out=FB, out_ch_offset=CA)
            FB.record_stream(main)      # allocated on the side stream, read (and released) on the caller's
            j1 = s1.record_event()
        with torch.cuda.stream(s2):
            s2.wait_event(fork)
            HD = conv2d(Rl, packs["HD"], CA, 3)
            HD.record_stream(main)
            j2 = s2.record_event()
        FA, HC = conv2d(L, packs["A"], CA, 3), conv2d(Rw, packs["HC"], CA, 3)
        main.wait_event(j1)
        main.wait_event(j2)
    else:
        FA, HC = conv2d(L, packs["A"], CA, 3), conv2d(Rw, packs["HC"], CA, 3)
        FB = torch.empty((B, 2 * CA, H, Wc), dtype=torch.float32, device=dev)
        conv2d(Lc, packs["B1"], CA, 3, out=FB, out_ch_offset=0)
        conv2d(Lc, packs["B2"], CA, 3, out=FB, out_ch_offset=CA)
        HD = conv2d(Rl, packs["HD"], CA, 3)
    FM = torch.empty((B, Co, H, W), dtype=torch.float32, device=dev)
    BAND = torch.empty((B, Co, H, D, 4), dtype=torch.float32, device=dev)
    GM = torch.empty((B, Co, H, W + 4), dtype=torch.float32, device=dev)
    GB = torch.empty((B, Co, H, D), dtype=torch.float32, device=dev)
    # (FB holds m = 1 in channels [0, CA) and m = 2 in [CA, 2 CA), each as dz * Co + co like FA)
    check(lib.dmb_catconv_finalize_f32(dev_ptr(FA), dev_ptr(FB), dev_ptr(HC), dev_ptr(HD), dev_ptr(FM), dev_ptr(BAND),
                                       dev_ptr(GM), dev_ptr(GB), B, Co, CA, 2 * CA, D, H, W, Wc, stream_ptr(dev)),
          "dmb_catconv_finalize_f32")
    out = torch.empty((B, Co, D, H, W), dtype=torch.float32, device=dev)
    check(lib.dmb_catconv_combine_f32(dev_ptr(FA), dev_ptr(HC), dev_ptr(FM), dev_ptr(BAND), dev_ptr(GM), dev_ptr(GB),
                                      dev_ptr(scale, allow_none=True), dev_ptr(shift, allow_none=True), dev_ptr(out), B, Co, CA,
                                      D, H, W, _relu_mode(relu), stream_ptr(dev)), "dmb_catconv_combine_f32")
    return out


# Opt-in arithmetic of the 32-channel stride-1 layers: "exact" (default) = FP32 MFMA, bitwise an fmaf chain;
# "bf16x6" = FP32 operands split exactly into 3 bf16 pieces, 6 cross products on the bf16 matrix cores, FP32 accumulate
# (EXPERIMENTAL: at least as accurate against FP64, not bit-identical; see csrc/conv3d_x6.hip).  Never changed implicitly.
_conv3d_mode = "exact"


def set_conv3d_mode(mode):
    global _conv3d_mode
    if mode not in ("exact", "bf16x6"):
        raise ValueError("conv3d mode must be 'exact' or 'bf16x6'")
    _conv3d_mode = mode


def conv3d_mode():
    return _conv3d_mode


def conv3d_x6_applicable(x, Co, stride):
    """The split kernel covers stride 1 with 32 output channels (W % 48 == 0) or 64 (W % 24 == 0)."""
    W = x.shape[-1]
    return stride == 1 and ((Co == 32 and W % 48 == 0) or (Co == 64 and W % 24 == 0)) and x.data_ptr() % 16 == 0


def pack_conv3d_x6_weights(w):
    lib = _lib.load()
    w = _f32c(w, "weight")
    Co, Ci = w.shape[0], w.shape[1]
    wp = torch.empty((lib.dmb_conv3d_x6_packed_bytes(Co, Ci) // 4,), dtype=torch.float32, device=w.device)
    check(lib.dmb_conv3d_x6_pack_weights_f32(dev_ptr(w), dev_ptr(wp), Co, Ci, stream_ptr(w.device)),
          "dmb_conv3d_x6_pack_weights_f32")
    return wp


def conv3d_k3_x6(x, wpack, Co, scale=None, shift=None, residual=None, relu=False):
    lib = _lib.load()
    x = _f32c(x, "x")
    B, Ci, D, H, W = x.shape
    y = torch.empty((B, Co, D, H, W), dtype=torch.float32, device=x.device)
    if residual is not None and tuple(residual.shape) != tuple(y.shape):
        raise _lib.DmbLibraryError("residual shape %s != output shape %s" % (tuple(residual.shape), tuple(y.shape)))
    tag = "conv3d_k3_s1_%dto%d" % (Ci, Co)
    if _kernel_timer is not None:
        _kernel_timer.start(tag)
    check(lib.dmb_conv3d_k3_x6_f32(dev_ptr(x), dev_ptr(wpack), dev_ptr(scale, allow_none=True),
                                   dev_ptr(shift, allow_none=True), dev_ptr(residual, allow_none=True), dev_ptr(y),
                                   B, Ci, Co, D, H, W, _relu_mode(relu), stream_ptr(x.device)), "dmb_conv3d_k3_x6_f32")
    if _kernel_timer is not None:
        _kernel_timer.stop(tag)
    return y


def conv3d_k3_c1(x, w, bias=0.0, residual=None):
    sh = _lib.shim()
    if sh is not None:
        return sh.conv3d_k3_c1(x if x.is_contiguous() else x.contiguous(), w if w.is_contiguous() else w.contiguous(), float(bias),
                               residual, _conv_flags())
    lib = _lib.load()
    x, w = _f32c(x, "x"), _f32c(w, "weight")
    B, Ci, D, H, W = x.shape
    y = torch.empty((B, 1, D, H, W), dtype=torch.float32, device=x.device)
    if w.numel() != Ci * 27:
        raise _lib.DmbLibraryError("conv3d_k3_c1: weight %s for %d input channels" % (tuple(w.shape), Ci))
    if residual is not None:
        _same_shape(residual, y, "conv3d_k3_c1 residual")
    check(lib.dmb_conv3d_k3_c1_f32(dev_ptr(x), dev_ptr(w), float(bias), dev_ptr(residual, allow_none=True), dev_ptr(y),
                                   B, Ci, D, H, W, _conv_flags(), stream_ptr(x.device)), "dmb_conv3d_k3_c1_f32")
    return y


# Work-item counters of the transposed convolution (include/dmb_hip.h: DMB_DECONV3D_WORKSPACE_BYTES): the library allocates
# nothing, so the host layer keeps ONE zeroed workspace per (device, stream) -- launches on one stream run one after the other
# and each leaves the workspace zeroed; launches on different streams may overlap and get different workspaces.  During a graph
# capture nothing is cached: a workspace allocated there comes from the graph's pool and its zero fill is a captured node -- it
# only holds zeros once THAT graph has been replayed, so every call inside a capture gets its own (one 3 us fill per call).
_deconv_ws = {}


def deconv3d_workspace(device):
    if torch.cuda.is_current_stream_capturing():
        return torch.zeros((_lib.DECONV3D_WORKSPACE_BYTES // 4,), dtype=torch.int32, device=device)
    st = torch.cuda.current_stream(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), st.cuda_stream)
    ws = _deconv_ws.get(key)
    if ws is None:
        ws = torch.zeros((_lib.DECONV3D_WORKSPACE_BYTES // 4,), dtype=torch.int32, device=device)
        _deconv_ws[key] = ws
    return ws


def deconv3d_k3s2(x, wpack, Co, scale=None, shift=None, residual=None, relu=False, workspace="auto", out=None, out_width=None):
    """``workspace``: "auto" = the per-stream workspace above; None = the kernel form without counters; or an int32 tensor of
    DECONV3D_WORKSPACE_BYTES holding zeros.  ``out_width``: for an input whose rows are zero-padded on the right to a multiple
    of 4 columns, the real output width (2 x the unpadded input width); default 2 W."""
    sh = _lib.shim()
    if sh is not None:
        if isinstance(workspace, str):
            workspace = deconv3d_workspace(x.device)
        return sh.deconv3d_k3s2(x if x.is_contiguous() else x.contiguous(), wpack, Co, scale, shift, residual,
                                _relu_mode(relu) | _conv_flags(), workspace, out, -1 if out_width is None else int(out_width))
    lib = _lib.load()
    x = _f32c(x, "x")
    B, Ci, D, H, W = x.shape
    Wout = 2 * W if out_width is None else int(out_width)
    y = _out_tensor(out, (B, Co, 2 * D, 2 * H, Wout), x.device, "deconv3d_k3s2")
    if residual is not None and tuple(residual.shape) != tuple(y.shape):
        raise _lib.DmbLibraryError("residual shape %s != output shape %s" % (tuple(residual.shape), tuple(y.shape)))
    _affine_ok(scale, shift, Co, "deconv3d_k3s2")
    if wpack.numel() != lib.dmb_deconv3d_packed_floats(Ci, Co):
        raise _lib.DmbLibraryError("deconv3d_k3s2: packed weights hold %d floats, %d -> %d channels need %d"
                                   % (wpack.numel(), Ci, Co, lib.dmb_deconv3d_packed_floats(Ci, Co)))
    if isinstance(workspace, str):
        workspace = deconv3d_workspace(x.device)
    if workspace is not None and (workspace.dtype != torch.int32 or workspace.numel() * 4 < _lib.DECONV3D_WORKSPACE_BYTES
                                  or workspace.device != x.device or not workspace.is_contiguous()):
        raise _lib.DmbLibraryError("deconv3d_k3s2: workspace must be a contiguous int32 tensor of %d bytes on %s"
                                   % (_lib.DECONV3D_WORKSPACE_BYTES, x.device))
    check(lib.dmb_deconv3d_k3s2_f32(dev_ptr(x), dev_ptr(wpack), dev_ptr(scale, allow_none=True),
                                    dev_ptr(shift, allow_none=True), dev_ptr(residual, allow_none=True), dev_ptr(y),
                                    B, Ci, Co, D, H, W, Wout, _relu_mode(relu) | _conv_flags(),
                                    None if workspace is None else ctypes.c_void_p(workspace.data_ptr()), stream_ptr(x.device)), "dmb_deconv3d_k3s2_f32")
    return y


# ---------------------------------------------------------------------------------------------- conv backward
# (backward passes of the convolution units.  "dc" = gradient w.r.t. the raw convolution output.)
def pack_conv3d_dgrad_weights(w):
    """Stride-1 nn.Conv3d weight [Co, Ci, 3, 3, 3] -> packed weights of its data-gradient convolution (Co -> Ci)."""
    lib = _lib.load()
    w = _f32c(w, "weight")
    Co, Ci = w.shape[0], w.shape[1]
    wp = torch.empty((lib.dmb_conv3d_packed_floats(Ci, Co),), dtype=torch.float32, device=w.device)
    check(lib.dmb_conv3d_pack_dgrad_weights_f32(dev_ptr(w), dev_ptr(wp), Co, Ci, stream_ptr(w.device)),
          "dmb_conv3d_pack_dgrad_weights_f32")
    return wp


class _PackJob(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p), ("wpack", ctypes.c_void_p), ("Co", ctypes.c_int), ("Ci", ctypes.c_int),
                ("mode", ctypes.c_int), ("reserved", ctypes.c_int)]


PACK_CONV, PACK_DECONV, PACK_DGRAD = 0, 1, 2   # dmb_pack_job.mode (include/dmb_hip.h)


def unit_pack_jobs(w, transposed, stride):
    """The two packs a training step needs of one convolution unit's weight, as (Co, Ci, mode) of dmb_pack_job: [forward, data
    gradient] (the table the reference's autograd hides in cuDNN: basic_layers.py:68-100,160-177 under train()).
      stride-1 Conv3d    [Co, Ci]: forward conv Ci -> Co; data gradient = conv Co -> Ci on mirrored taps
      stride-2 Conv3d    [Co, Ci]: forward conv; data gradient = the transposed convolution with the same tensor ([Ci_t, Co_t] = [Co, Ci])
      ConvTranspose3d    [Ci, Co]: forward transposed; data gradient = stride-2 conv reading the tensor as [Co', Ci'] = [Ci, Co]"""
    a, b = int(w.shape[0]), int(w.shape[1])
    if transposed:
        return [(b, a, PACK_DECONV), (a, b, PACK_CONV)]
    if stride == 1:
        return [(a, b, PACK_CONV), (b, a, PACK_DGRAD)]
    return [(a, b, PACK_CONV), (b, a, PACK_DECONV)]


def packed_floats(Co, Ci):
    return int(_lib.load().dmb_conv3d_packed_floats(Co, Ci))


def make_pack_table(jobs, device):
    """jobs: list of (w, wpack, Co, Ci, mode) with device tensors -> the dmb_pack_job table as a device tensor (keep it, and the
    tensors it points to, alive while it is used)."""
    arr = (_PackJob * len(jobs))()
    for i, (w, wp, Co, Ci, mode) in enumerate(jobs):
        if not (w.is_cuda and wp.is_cuda and w.is_contiguous() and wp.is_contiguous() and w.dtype == torch.float32 and wp.dtype == torch.float32):
            raise _lib.DmbLibraryError("make_pack_table: contiguous float32 device tensors expected")
        if wp.numel() != packed_floats(Co, Ci) or w.numel() != Co * Ci * 27:
            raise _lib.DmbLibraryError("make_pack_table: job %d sizes do not fit %d -> %d channels" % (i, Ci, Co))
        arr[i] = _PackJob(w.data_ptr(), wp.data_ptr(), Co, Ci, mode, 0)
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    return host.to(device)


def run_pack_table(table, njobs):
    lib = _lib.load()
    check(lib.dmb_conv3d_pack_weights_multi_f32(ctypes.c_void_p(table.data_ptr()), int(njobs), stream_ptr(table.device)),
          "dmb_conv3d_pack_weights_multi_f32")


def conv3d_k3_dgrad(dc, w, stride=1, in_size=None, residual=None, wpack=None):
    """Gradient of nn.Conv3d(k=3, padding=1, stride) w.r.t. its input; w is the layer's weight [Co, Ci, 3, 3, 3].
    ``in_size`` = (D, H, W) of that input; needed for stride 2 when an extent is odd (the adjoint is computed for the even
    size 2 x output and its last plane / row / column dropped).  ``residual``: a tensor of the input's shape added to the result
    in the kernel's epilogue (the gradient the input already holds from its other consumers: train_fn's gradient carry).
    ``wpack``: the data gradient's packed weights when the caller already holds them (unit_pack_jobs)."""
    Co, Ci = w.shape[0], w.shape[1]
    if stride == 1:
        return conv3d_k3(dc, wpack if wpack is not None else pack_conv3d_dgrad_weights(w), Ci, residual=residual)
    if stride == 2:
        # the adjoint of a stride-2 convolution is the transposed convolution with the same weight tensor; the transposed kernel
        # takes 64 or <= 32 output channels per launch, so wider inputs (GC-Net: 96, 128) are done in channel chunks
        full = tuple(2 * e for e in dc.shape[2:])
        fused = residual is not None and (Ci == 64 or Ci <= 32) and (in_size is None or tuple(in_size) == full)
        if Ci == 64 or Ci <= 32:
            dx = deconv3d_k3s2(dc, wpack if wpack is not None else pack_deconv3d_weights(w), Ci, residual=residual if fused else None)
        else:
            parts, c0 = [], 0
            while c0 < Ci:
                n = 64 if Ci - c0 >= 64 else min(32, Ci - c0)
                parts.append(deconv3d_k3s2(dc, pack_deconv3d_weights(w[:, c0:c0 + n].contiguous()), n))
                c0 += n
            dx = torch.cat(parts, 1)
        if in_size is not None and tuple(in_size) != tuple(dx.shape[2:]):
            D, H, W = in_size
            if any(a not in (b, b - 1) for a, b in zip((D, H, W), dx.shape[2:])):
                raise _lib.DmbLibraryError("conv3d_k3_dgrad: input size %s does not belong to output size %s" % (tuple(in_size), tuple(dc.shape[2:])))
            dx = dx[:, :, :D, :H, :W].contiguous()
        if residual is not None and not fused:
            dx = dx + residual
        return dx
    raise _lib.DmbLibraryError("conv3d_k3_dgrad: stride must be 1 or 2")


def deconv3d_k3s2_dgrad(dy, w, residual=None, wpack=None):
    """Gradient of nn.ConvTranspose3d(k=3, s=2, p=1, output_padding=1) w.r.t. its input; w is [Ci, Co, 3, 3, 3].  ``residual`` and
    ``wpack`` as in conv3d_k3_dgrad."""
    return conv3d_k3(dy, wpack if wpack is not None else pack_conv3d_weights(w), w.shape[0], residual=residual, stride=2)


def conv3d_k3_wgrad(x, dc):
    """Gradient of a stride-1 nn.Conv3d(k=3, padding=1) w.r.t. its weight: [Co, Ci, 3, 3, 3]."""
    lib = _lib.load()
    x, dc = _f32c(x, "x"), _f32c(dc, "dc")
    B, Ci, D, H, W = x.shape
    Co = dc.shape[1]
    if tuple(dc.shape) != (B, Co, D, H, W):
        raise _lib.DmbLibraryError("conv3d_k3_wgrad: dc shape %s does not match x %s" % (tuple(dc.shape), tuple(x.shape)))
    dw = torch.empty((Co, Ci, 3, 3, 3), dtype=torch.float32, device=x.device)
    ws = torch.empty((lib.dmb_conv3d_wgrad_workspace_floats(Co, Ci),), dtype=torch.float32, device=x.device)
    check(lib.dmb_conv3d_k3_wgrad_f32(dev_ptr(x), dev_ptr(dc), dev_ptr(dw), dev_ptr(ws), B, Ci, Co, D, H, W,
                                      stream_ptr(x.device)), "dmb_conv3d_k3_wgrad_f32")
    return dw


def _s2_wgrad(small, big, what):
    lib = _lib.load()
    small, big = _f32c(small, "small"), _f32c(big, "big")
    B, Cs, Ds, Hs, Ws = small.shape
    Bb, Cb, Db, Hb, Wb = big.shape
    if Bb != B:
        raise _lib.DmbLibraryError("%s: batch sizes differ" % what)
    dw = torch.empty((Cs, Cb, 3, 3, 3), dtype=torch.float32, device=small.device)
    ws = torch.empty((lib.dmb_conv3d_wgrad_workspace_floats(Cs, Cb),), dtype=torch.float32, device=small.device)
    check(lib.dmb_conv3d_k3s2_wgrad_f32(dev_ptr(small), dev_ptr(big), dev_ptr(dw), dev_ptr(ws), B, Cs, Cb, Ds, Hs, Ws, Db, Hb, Wb,
                                        stream_ptr(small.device)), what)
    return dw


def conv3d_k3s2_wgrad(x, dc):
    """Weight gradient of nn.Conv3d(k=3, stride=2, padding=1): [Co, Ci, 3, 3, 3]."""
    return _s2_wgrad(dc, x, "dmb_conv3d_k3s2_wgrad_f32 (conv)")


def deconv3d_k3s2_wgrad(x, dy):
    """Weight gradient of nn.ConvTranspose3d(k=3, s=2, p=1, output_padding=1): [Ci, Co, 3, 3, 3]."""
    return _s2_wgrad(x, dy, "dmb_conv3d_k3s2_wgrad_f32 (transposed conv)")


def conv2d_wgrad(x, dc, ksize=3, dilation=1):
    """Weight gradient of a stride-1 nn.Conv2d (k = 1, or k = 3 with dilation 1 | 2, padding = dilation * (k // 2)):
    x [B, Ci, H, W], dc [B, Co, H, W] -> [Co, Ci, k, k]."""
    lib = _lib.load()
    x, dc = _f32c(x, "x"), _f32c(dc, "dc")
    B, Ci, H, W = x.shape
    Co = dc.shape[1]
    if tuple(dc.shape) != (B, Co, H, W):
        raise _lib.DmbLibraryError("conv2d_wgrad: dc shape %s does not match x %s" % (tuple(dc.shape), tuple(x.shape)))
    if W % 4:
        # the kernel stages 16-byte units: zero columns on the right change nothing (dc is zero there, x reads as padding)
        pad = 4 - W % 4
        x, dc = torch.nn.functional.pad(x, (0, pad)), torch.nn.functional.pad(dc, (0, pad))
        W += pad
    sh = _lib.shim()
    if sh is not None:
        return sh.conv2d_wgrad(x, dc, int(ksize), int(dilation))
    dw = torch.empty((Co, Ci, ksize, ksize), dtype=torch.float32, device=x.device)
    ws = torch.empty((lib.dmb_conv2d_wgrad_workspace_floats(Co, Ci),), dtype=torch.float32, device=x.device)
    check(lib.dmb_conv2d_wgrad_f32(dev_ptr(x), dev_ptr(dc), dev_ptr(dw), dev_ptr(ws), B, Ci, Co, H, W, int(ksize), int(dilation),
                                   stream_ptr(x.device)), "dmb_conv2d_wgrad_f32")
    return dw


def conv2d_k3_wgrad(x, dc):
    return conv2d_wgrad(x, dc, 3, 1)


def conv2d_dgrad_packs(w):
    """Packed weights of the data-gradient convolution of a stride-1 nn.Conv2d with weight [Co, Ci, k, k]: the layer's taps mirrored,
    channel roles exchanged, in chunks of at most 128 output (= the layer's input) channels: [(c0, n, pack), ...]."""
    w = _f32c(w, "weight")
    Ci = w.shape[1]
    wt = w.detach().transpose(0, 1).flip(2, 3).contiguous()           # [Ci, Co, k, k]: a convolution Co -> Ci
    out = []
    for c0 in range(0, Ci, 128):
        n = min(128, Ci - c0)
        out.append((c0, n, pack_conv2d_weights(wt[c0:c0 + n].contiguous() if n != Ci else wt)))
    return out


def conv2d_dgrad(dc, w, dilation=1, residual=None, packs=None):
    """Gradient of a stride-1 nn.Conv2d (k in {1, 3}, padding = dilation * (k // 2)) w.r.t. its input; w is the layer's weight
    [Co, Ci, k, k].  The same convolution kernel on mirrored, channel-exchanged weights, at most 128 output channels per launch.
    ``residual`` ([B, Ci, H, W]) is added in the epilogue (conv3d_k3_dgrad); ``packs``: conv2d_dgrad_packs(w) when the caller holds
    them already."""
    Ci, k = w.shape[1], w.shape[2]
    dc = _f32c(dc, "dc")
    B, _, H, W = dc.shape
    dx = torch.empty((B, Ci, H, W), dtype=torch.float32, device=dc.device)
    if residual is not None:
        residual = _f32c(residual, "residual")
    for c0, n, pack in (packs if packs is not None else conv2d_dgrad_packs(w)):
        conv2d(dc, pack, n, k, dilation=dilation, residual=residual, out=dx, out_ch_offset=c0, res_ch_offset=c0)
    return dx


# ---------------------------------------------------------------------------------------------- BatchNorm (training)
def _bcs(t):
    """[B, C, *spatial] -> (B, C, S)."""
    S = 1
    for d in t.shape[2:]:
        S *= int(d)
    return int(t.shape[0]), int(t.shape[1]), S


def bn_train_stats(c, gamma=None, beta=None, running_mean=None, running_var=None, momentum=0.1, eps=1e-5):
    """Batch statistics of a raw convolution output; returns (mean, invstd, scale, shift) and updates the running buffers."""
    lib = _lib.load()
    c = _f32c(c, "c")
    B, C, S = _bcs(c)
    out = [torch.empty((C,), dtype=torch.float32, device=c.device) for _ in range(4)]
    ws = torch.empty((lib.dmb_bn_workspace_doubles(C, S),), dtype=torch.float64, device=c.device)
    check(lib.dmb_bn_train_stats_f32(dev_ptr(c), dev_ptr(gamma, allow_none=True), dev_ptr(beta, allow_none=True),
                                     dev_ptr(running_mean, allow_none=True), dev_ptr(running_var, allow_none=True),
                                     float(momentum), float(eps), dev_ptr(out[0]), dev_ptr(out[1]), dev_ptr(out[2]),
                                     dev_ptr(out[3]), dev_ptr(ws), B, C, S, stream_ptr(c.device)), "dmb_bn_train_stats_f32")
    return tuple(out)


def conv3d_k3_bnstats(x, wpack, Co):
    """The RAW stride-1 convolution of a training-mode unit + the partial sums of its batch statistics from the kernel's epilogue
    (dmb_conv3d_k3_bnstats_f32): returns (raw, partials [Co, P, 2] float64), or None where the shape is not covered (the caller
    then runs conv3d_k3 and bn_train_fwd)."""
    lib = _lib.load()
    x = _f32c(x, "x")
    B, Ci, D, H, W = x.shape
    P = int(lib.dmb_conv3d_k3_bnstats_partials(B, Ci, Co, D, H, W))
    if P <= 0 or x.data_ptr() % 16:
        return None
    if wpack.numel() != lib.dmb_conv3d_packed_floats(Co, Ci):
        raise _lib.DmbLibraryError("conv3d_k3_bnstats: packed weights hold %d floats, %d -> %d channels need %d"
                                   % (wpack.numel(), Ci, Co, lib.dmb_conv3d_packed_floats(Co, Ci)))
    raw = torch.empty((B, Co, D, H, W), dtype=torch.float32, device=x.device)
    parts = torch.empty((Co, P, 2), dtype=torch.float64, device=x.device)
    check(lib.dmb_conv3d_k3_bnstats_f32(dev_ptr(x), dev_ptr(wpack), dev_ptr(raw), dev_ptr(parts), B, Ci, Co, D, H, W, stream_ptr(x.device)),
          "dmb_conv3d_k3_bnstats_f32")
    return raw, parts


def bn_train_act(c, partials, gamma=None, beta=None, running_mean=None, running_var=None, num_batches_tracked=None, momentum=0.1, eps=1e-5,
                 residual=None, relu=False):
    """bn_train_fwd without its pass over ``c`` for the block sums: the statistics are finished from ``partials`` ([C, P, 2] float64
    sums of c and c^2, conv3d_k3_bnstats).  Returns (y, mean, invstd, scale, shift)."""
    lib = _lib.load()
    c = _f32c(c, "c")
    B, C, S = _bcs(c)
    if partials.dtype != torch.float64 or partials.dim() != 3 or partials.shape[0] != C or partials.shape[2] != 2 or not partials.is_contiguous():
        raise _lib.DmbLibraryError("bn_train_act: partials must be a contiguous float64 [C, P, 2] tensor")
    if residual is not None and (tuple(residual.shape) != tuple(c.shape) or not residual.is_contiguous() or residual.dtype != torch.float32):
        raise _lib.DmbLibraryError("bn_train_act: residual must be a contiguous float32 tensor of shape %s" % (tuple(c.shape),))
    if num_batches_tracked is not None and (num_batches_tracked.dtype != torch.int64 or num_batches_tracked.device != c.device):
        raise _lib.DmbLibraryError("bn_train_act: num_batches_tracked must be an int64 tensor on %s" % c.device)
    y = torch.empty_like(c)
    stats = torch.empty((4, C), dtype=torch.float32, device=c.device)
    check(lib.dmb_bn_train_act_f32(dev_ptr(c), dev_ptr(partials), int(partials.shape[1]), dev_ptr(gamma, allow_none=True),
                                   dev_ptr(beta, allow_none=True), dev_ptr(running_mean, allow_none=True),
                                   dev_ptr(running_var, allow_none=True), dev_ptr(num_batches_tracked, allow_none=True),
                                   float(momentum), float(eps), dev_ptr(stats[0]), dev_ptr(stats[1]), dev_ptr(stats[2]), dev_ptr(stats[3]),
                                   dev_ptr(residual, allow_none=True), dev_ptr(y), B, C, S, _relu_mode(relu), stream_ptr(c.device)),
          "dmb_bn_train_act_f32")
    written = [t for t in (running_mean, running_var, num_batches_tracked) if t is not None]
    if written:
        torch.autograd.graph.increment_version(written)
    return y, stats[0], stats[1], stats[2], stats[3]


def bn_act(c, scale, shift, residual=None, relu=False):
    """y = act(c*scale + shift (+ residual)); relu as in conv3d_k3 (False / True / 'pre')."""
    lib = _lib.load()
    c = _f32c(c, "c")
    B, C, S = _bcs(c)
    y = torch.empty_like(c)
    check(lib.dmb_bn_act_f32(dev_ptr(c), dev_ptr(scale), dev_ptr(shift), dev_ptr(residual, allow_none=True), dev_ptr(y),
                             B, C, S, _relu_mode(relu), stream_ptr(c.device)), "dmb_bn_act_f32")
    return y


def bn_train_fwd(c, gamma=None, beta=None, running_mean=None, running_var=None, num_batches_tracked=None, momentum=0.1, eps=1e-5,
                 residual=None, relu=False):
    """bn_train_stats + bn_act of a batch-statistics unit in two launches (dmb_bn_train_fwd_f32): returns
    (y, mean, invstd, scale, shift); updates the running buffers and adds 1 to ``num_batches_tracked`` (an int64 device tensor)."""
    sh = _lib.shim()
    if sh is not None:      # the torch-extension shim: the same checks and the same C-ABI call without the interpreter
        y, stats = sh.bn_train_fwd(c if c.is_contiguous() else c.contiguous(), gamma, beta, running_mean, running_var, num_batches_tracked,
                                   float(momentum), float(eps), residual, _relu_mode(relu))
        return y, stats[0], stats[1], stats[2], stats[3]
    lib = _lib.load()
    c = _f32c(c, "c")
    B, C, S = _bcs(c)
    y = torch.empty_like(c)
    if residual is not None and (tuple(residual.shape) != tuple(c.shape) or not residual.is_contiguous() or residual.dtype != torch.float32):
        raise _lib.DmbLibraryError("bn_train_fwd: residual must be a contiguous float32 tensor of shape %s" % (tuple(c.shape),))
    if num_batches_tracked is not None and (num_batches_tracked.dtype != torch.int64 or num_batches_tracked.device != c.device):
        raise _lib.DmbLibraryError("bn_train_fwd: num_batches_tracked must be an int64 tensor on %s" % c.device)
    stats = torch.empty((4, C), dtype=torch.float32, device=c.device)
    ws = torch.empty((lib.dmb_bn_workspace_doubles(C, S),), dtype=torch.float64, device=c.device)
    check(lib.dmb_bn_train_fwd_f32(dev_ptr(c), dev_ptr(gamma, allow_none=True), dev_ptr(beta, allow_none=True),
                                   dev_ptr(running_mean, allow_none=True), dev_ptr(running_var, allow_none=True),
                                   None if num_batches_tracked is None else ctypes.c_void_p(num_batches_tracked.data_ptr()),
                                   float(momentum), float(eps), dev_ptr(stats[0]), dev_ptr(stats[1]), dev_ptr(stats[2]),
                                   dev_ptr(stats[3]), dev_ptr(residual, allow_none=True), dev_ptr(y), dev_ptr(ws), B, C, S,
                                   _relu_mode(relu), stream_ptr(c.device)), "dmb_bn_train_fwd_f32")
    # the kernel wrote these through their pointers: tell torch (the modules' folded-parameter caches key on the versions --
    # rounds 1-5 relied on the ``num_batches_tracked += 1`` launch for that)
    written = [t for t in (running_mean, running_var, num_batches_tracked) if t is not None]
    if written:
        torch.autograd.graph.increment_version(written)
    return y, stats[0], stats[1], stats[2], stats[3]


def bn_act_bwd(dy, c, y, scale, shift, mean, invstd, relu=False, training=True, want_dres=False, dres_acc=None):
    """Backward of bn_act (+ the batch statistics if training): returns (dc, dgamma, dbeta, dres or None).  ``dres_acc``: a
    gradient the skip operand already holds, added into ``dres`` by the same pass (implies ``want_dres``)."""
    sh = _lib.shim()
    if sh is not None:
        dc, gb, dres = sh.bn_act_bwd(dy if dy.is_contiguous() else dy.contiguous(), c if c.is_contiguous() else c.contiguous(), y, scale, shift,
                                     mean, invstd, _relu_mode(relu), bool(training), bool(want_dres),
                                     None if dres_acc is None else (dres_acc if dres_acc.is_contiguous() else dres_acc.contiguous()))
        return dc, gb[0], gb[1], dres
    lib = _lib.load()
    dy, c = _f32c(dy, "dy"), _f32c(c, "c")
    B, C, S = _bcs(c)
    mode = _relu_mode(relu)
    dc = torch.empty_like(c)
    if dres_acc is not None:
        dres_acc = _f32c(dres_acc, "dres_acc")
        if tuple(dres_acc.shape) != tuple(c.shape):
            raise _lib.DmbLibraryError("bn_act_bwd: dres_acc shape %s != %s" % (tuple(dres_acc.shape), tuple(c.shape)))
        want_dres = True
    dres = torch.empty_like(c) if want_dres else None
    gb = torch.empty((2, C), dtype=torch.float32, device=c.device)
    ws = torch.empty((lib.dmb_bn_workspace_doubles(C, S),), dtype=torch.float64, device=c.device)
    check(lib.dmb_bn_act_bwd_f32(dev_ptr(dy), dev_ptr(c), dev_ptr(y, allow_none=mode != 1), dev_ptr(scale), dev_ptr(shift),
                                 dev_ptr(mean), dev_ptr(invstd), dev_ptr(ws), dev_ptr(gb[0]), dev_ptr(gb[1]), dev_ptr(dc),
                                 dev_ptr(dres, allow_none=True), dev_ptr(dres_acc, allow_none=True), B, C, S, mode,
                                 1 if training else 0, stream_ptr(c.device)), "dmb_bn_act_bwd_f32")
    return dc, gb[0], gb[1], dres


def channel_dot(a, g):
    """sum over (batch, space) of a[b, c, s] * g[b, 0, s] -> [C]."""
    lib = _lib.load()
    a, g = _f32c(a, "a"), _f32c(g, "g")
    B, C, S = _bcs(a)
    if g.shape[0] != B or g.numel() != B * S:
        raise _lib.DmbLibraryError("channel_dot: g shape %s does not match a %s" % (tuple(g.shape), tuple(a.shape)))
    out = torch.empty((C,), dtype=torch.float32, device=a.device)
    ws = torch.empty((lib.dmb_bn_workspace_doubles(C, S),), dtype=torch.float64, device=a.device)
    check(lib.dmb_channel_dot_f32(dev_ptr(a), dev_ptr(g), dev_ptr(ws), dev_ptr(out), B, C, S, stream_ptr(a.device)), "dmb_channel_dot_f32")
    return out


# ---------------------------------------------------------------------------------------------- path ends, backward
def _volume_bwd(fn_name, dvol, disp_idx, C):
    lib = _lib.load()
    dvol = _f32c(dvol, "dvol")
    B, VC, D, H, W = dvol.shape
    if D != len(disp_idx) or VC != (2 * C if fn_name == "dmb_cat_fms_bwd_f32" else C):
        raise _lib.DmbLibraryError("%s: gradient shape %s does not match C=%d, D=%d" % (fn_name, tuple(dvol.shape), C, len(disp_idx)))
    dL = torch.empty((B, C, H, W), dtype=torch.float32, device=dvol.device)
    dR = torch.empty_like(dL)
    check(getattr(lib, fn_name)(dev_ptr(dvol), dev_ptr(dL), dev_ptr(dR), B, C, H, W, D, host_ints(disp_idx),
                                stream_ptr(dvol.device)), fn_name)
    return dL, dR


def cat_first_wgrad(left, right, dc, kind="cat"):
    """Weight gradient [Co, 2C (cat) | C (dif), 3, 3, 3] of the first convolution on the volume of cat_fms / dif_fms(left, right) with
    unit disparity step, from dc [B, Co, D, H, W], without the volume: dmb_cat_first_wgrad_maps_f32 + two 2-D weight gradients."""
    lib = _lib.load()
    left, right, dc = _f32c(left, "left"), _f32c(right, "right"), _f32c(dc, "dc")
    B, Co, D, H, W = dc.shape
    C = left.shape[1]
    if tuple(left.shape) != (B, C, H, W) or tuple(right.shape) != (B, C, H, W):
        raise _lib.DmbLibraryError("cat_first_wgrad: feature maps %s / %s do not belong to dc %s" % (tuple(left.shape), tuple(right.shape), tuple(dc.shape)))
    maps = torch.empty((2, B, 9 * Co, H, W), dtype=torch.float32, device=dc.device)
    check(lib.dmb_cat_first_wgrad_maps_f32(dev_ptr(dc), dev_ptr(maps[0]), dev_ptr(maps[1]), B, Co, D, H, W, stream_ptr(dc.device)),
          "dmb_cat_first_wgrad_maps_f32")
    dl = conv2d_wgrad(left, maps[0], 3, 1).view(3, 3, Co, C, 3, 3)     # [dz, dx, co, ci, dy', dx']
    dr = conv2d_wgrad(right, maps[1], 3, 1).view(3, 3, Co, C, 3, 3)
    wl = torch.diagonal(dl, dim1=1, dim2=5).permute(1, 2, 0, 3, 4)      # tap dx' = dx  -> [co, ci, dz, dy, dx]
    wr = dr[..., 1].permute(2, 3, 0, 4, 1)                              # no column shift on the right half
    return (wl - wr).contiguous() if kind == "dif" else torch.cat([wl, wr], 1).contiguous()


def cat_fms_bwd(dvol, disp_idx):
    """Gradient of cat_fms w.r.t. (reference_fm, target_fm); dvol [B, 2C, D, H, W]."""
    return _volume_bwd("dmb_cat_fms_bwd_f32", dvol, disp_idx, dvol.shape[1] // 2)


def dif_fms_bwd(dvol, disp_idx):
    return _volume_bwd("dmb_dif_fms_bwd_f32", dvol, disp_idx, dvol.shape[1])


def soft_argmin_bwd(cost, disp, grad_disp, disp_values, alpha=1.0):
    lib = _lib.load()
    cost, disp, grad_disp = _f32c(cost, "cost"), _f32c(disp, "disp"), _f32c(grad_disp, "grad_disp")
    B, D, H, W = cost.shape
    out = torch.empty_like(cost)
    check(lib.dmb_soft_argmin_bwd_f32(dev_ptr(cost), dev_ptr(disp), dev_ptr(grad_disp), dev_ptr(out), B, D, H, W,
                                      float(alpha), host_floats(disp_values), stream_ptr(cost.device)), "dmb_soft_argmin_bwd_f32")
    return out


def trilinear_ac_soft_argmin_bwd(x, disp, grad_disp, out_size, disp_values, alpha=1.0, grad_cost=None):
    """Gradient of trilinear_ac_soft_argmin w.r.t. the low-resolution cost x [B, Di, Hi, Wi]: through the disparity
    (grad_disp) and, if given, through the up-sampled volume itself (grad_cost [B, Do, Ho, Wo])."""
    lib = _lib.load()
    x, disp, grad_disp = _f32c(x, "x"), _f32c(disp, "disp"), _f32c(grad_disp, "grad_disp")
    B, Di, Hi, Wi = x.shape
    Do, Ho, Wo = out_size
    if grad_cost is not None:
        grad_cost = _f32c(grad_cost, "grad_cost")
        if tuple(grad_cost.shape) != (B, Do, Ho, Wo):
            raise _lib.DmbLibraryError("trilinear_ac_soft_argmin_bwd: grad_cost shape %s" % (tuple(grad_cost.shape),))
    scratch = torch.empty((B, Di, Ho, Wo), dtype=torch.float32, device=x.device)
    gx = torch.empty_like(x)
    check(lib.dmb_trilinear_ac_soft_argmin_bwd_f32(dev_ptr(x), dev_ptr(disp), dev_ptr(grad_disp), dev_ptr(grad_cost, allow_none=True),
                                                   dev_ptr(scratch), dev_ptr(gx), B, Di, Hi, Wi, Do, Ho, Wo, float(alpha),
                                                   host_floats(disp_values), stream_ptr(x.device)), "dmb_trilinear_ac_soft_argmin_bwd_f32")
    return gx


def trilinear_ac_bwd(grad_y, in_size):
    """Gradient of trilinear_ac w.r.t. its input: grad_y [B, Do, Ho, Wo] -> [B, Di, Hi, Wi]."""
    lib = _lib.load()
    grad_y = _f32c(grad_y, "grad_y")
    B, Do, Ho, Wo = grad_y.shape
    Di, Hi, Wi = in_size
    scratch = torch.empty((B, Di, Ho, Wo), dtype=torch.float32, device=grad_y.device)
    gx = torch.empty((B, Di, Hi, Wi), dtype=torch.float32, device=grad_y.device)
    check(lib.dmb_trilinear_ac_bwd_f32(dev_ptr(grad_y), dev_ptr(scratch), dev_ptr(gx), B, Di, Hi, Wi, Do, Ho, Wo,
                                       stream_ptr(grad_y.device)), "dmb_trilinear_ac_bwd_f32")
    return gx


def avgpool2d_bwd(grad_y, in_hw, k):
    lib = _lib.load()
    grad_y = _f32c(grad_y, "grad_y")
    B, C = grad_y.shape[0], grad_y.shape[1]
    H, W = in_hw
    gx = torch.empty((B, C, H, W), dtype=torch.float32, device=grad_y.device)
    check(lib.dmb_avgpool2d_bwd_f32(dev_ptr(grad_y), dev_ptr(gx), B, C, H, W, int(k), stream_ptr(grad_y.device)), "dmb_avgpool2d_bwd_f32")
    return gx


def bilinear_ac_bwd(grad_y, in_hw):
    lib = _lib.load()
    grad_y = _f32c(grad_y, "grad_y")
    B, C, Ho, Wo = grad_y.shape
    Hi, Wi = in_hw
    gx = torch.empty((B, C, Hi, Wi), dtype=torch.float32, device=grad_y.device)
    check(lib.dmb_bilinear_ac_bwd_f32(dev_ptr(grad_y), dev_ptr(gx), B, C, Hi, Wi, Ho, Wo, stream_ptr(grad_y.device)), "dmb_bilinear_ac_bwd_f32")
    return gx


def bilinear_scale_bwd(grad_y, in_hw, mult):
    lib = _lib.load()
    grad_y = _f32c(grad_y, "grad_y")
    B, C, Ho, Wo = grad_y.shape
    Hi, Wi = in_hw
    gx = torch.empty((B, C, Hi, Wi), dtype=torch.float32, device=grad_y.device)
    check(lib.dmb_bilinear_scale_bwd_f32(dev_ptr(grad_y), dev_ptr(gx), B, C, Hi, Wi, Ho, Wo, float(mult), stream_ptr(grad_y.device)),
          "dmb_bilinear_scale_bwd_f32")
    return gx


def deconv3d_k8s4_c1_bwd(x, w, dy, want_dx=True, want_dw=True):
    """Backward of deconv3d_k8s4_c1: (dx [B, D, H, W] or None, dw [8, 8, 8] or None)."""
    lib = _lib.load()
    x, w, dy = _f32c(x, "x"), _f32c(w, "weight"), _f32c(dy, "dy")
    B, D, H, W = x.shape
    if tuple(dy.shape) != (B, 4 * D, 4 * H, 4 * W):
        raise _lib.DmbLibraryError("deconv3d_k8s4_c1_bwd: dy shape %s does not match x %s" % (tuple(dy.shape), tuple(x.shape)))
    dx = torch.empty_like(x) if want_dx else None
    dw = torch.empty((8, 8, 8), dtype=torch.float32, device=x.device) if want_dw else None
    ws = torch.empty((lib.dmb_deconv3d_k8s4_bwd_workspace_doubles(),), dtype=torch.float64, device=x.device) if want_dw else None
    check(lib.dmb_deconv3d_k8s4_c1_bwd_f32(dev_ptr(x), dev_ptr(w), dev_ptr(dy), dev_ptr(dx, allow_none=True),
                                           dev_ptr(dw, allow_none=True), dev_ptr(ws, allow_none=True), B, D, H, W,
                                           stream_ptr(x.device)), "dmb_deconv3d_k8s4_c1_bwd_f32")
    return dx, dw


# ---------------------------------------------------------------------------------------------- upsampling
def trilinear_ac(x, out_size):
    """x: [B, Di, Hi, Wi] (single channel squeezed) -> [B, Do, Ho, Wo], align_corners=True."""
    lib = _lib.load()
    x = _f32c(x, "x")
    B, Di, Hi, Wi = x.shape
    Do, Ho, Wo = out_size
    y = torch.empty((B, Do, Ho, Wo), dtype=torch.float32, device=x.device)
    check(lib.dmb_trilinear_ac_f32(dev_ptr(x), dev_ptr(y), B, Di, Hi, Wi, Do, Ho, Wo, stream_ptr(x.device)),
          "dmb_trilinear_ac_f32")
    return y


def deconv3d_k8s4_c1(x, w):
    """x: [B, D, H, W], w: [8, 8, 8] (ConvTranspose3d(1,1,8,4,2) weight squeezed) -> [B, 4D, 4H, 4W]."""
    lib = _lib.load()
    x, w = _f32c(x, "x"), _f32c(w, "weight")
    B, D, H, W = x.shape
    y = torch.empty((B, 4 * D, 4 * H, 4 * W), dtype=torch.float32, device=x.device)
    check(lib.dmb_deconv3d_k8s4_c1_f32(dev_ptr(x), dev_ptr(w), dev_ptr(y), B, D, H, W, stream_ptr(x.device)),
          "dmb_deconv3d_k8s4_c1_f32")
    return y


# ---------------------------------------------------------------------------------------------- regression
def soft_argmin(cost, disp_values, alpha=1.0, normalize=True):
    lib = _lib.load()
    cost = _f32c(cost, "cost_volume")
    B, D, H, W = cost.shape
    if len(disp_values) != D:
        raise _lib.DmbLibraryError("The number of disparity samples should be consistent!")
    disp = torch.empty((B, 1, H, W), dtype=torch.float32, device=cost.device)
    check(lib.dmb_soft_argmin_f32(dev_ptr(cost), dev_ptr(disp), B, D, H, W, float(alpha), int(bool(normalize)),
                                  host_floats(disp_values), stream_ptr(cost.device)), "dmb_soft_argmin_f32")
    return disp


def soft_argmin_sampled(cost, disp_sample, alpha=1.0, normalize=True):
    lib = _lib.load()
    cost, disp_sample = _f32c(cost, "cost_volume"), _f32c(disp_sample, "disp_sample")
    B, D, H, W = cost.shape
    _same_shape(disp_sample, cost, "soft_argmin: cost volume and per-pixel disparity samples")
    disp = torch.empty((B, 1, H, W), dtype=torch.float32, device=cost.device)
    check(lib.dmb_soft_argmin_sampled_f32(dev_ptr(cost), dev_ptr(disp_sample), dev_ptr(disp), B, D, H, W, float(alpha),
                                          int(bool(normalize)), stream_ptr(cost.device)), "dmb_soft_argmin_sampled_f32")
    return disp


def local_soft_argmin(cost, radius, radius_dilation=1, start_disp=0, dilation=1, alpha=1.0, return_index=False):
    lib = _lib.load()
    cost = _f32c(cost, "cost_volume")
    B, D, H, W = cost.shape
    disp = torch.empty((B, 1, H, W), dtype=torch.float32, device=cost.device)
    idx = torch.empty((B, 1, H, W), dtype=torch.int64, device=cost.device) if return_index else None
    check(lib.dmb_local_soft_argmin_f32(dev_ptr(cost), dev_ptr(disp), dev_ptr(idx, allow_none=True), B, D, H, W,
                                        int(radius), int(radius_dilation), int(start_disp), int(dilation), float(alpha),
                                        stream_ptr(cost.device)), "dmb_local_soft_argmin_f32")
    return (disp, idx) if return_index else disp


def trilinear_soft_argmin(x, out_size, disp_values, alpha=1.0):
    lib = _lib.load()
    x = _f32c(x, "x")
    B, Di, Hi, Wi = x.shape
    Do, Ho, Wo = out_size
    disp = torch.empty((B, 1, Ho, Wo), dtype=torch.float32, device=x.device)
    check(lib.dmb_trilinear_soft_argmin_f32(dev_ptr(x), dev_ptr(disp), B, Di, Hi, Wi, Do, Ho, Wo, float(alpha),
                                            host_floats(disp_values), stream_ptr(x.device)),
          "dmb_trilinear_soft_argmin_f32")
    return disp


def trilinear_ac_soft_argmin(x, out_size, disp_values, alpha=1.0):
    """Up-sampled cost volume AND its soft-argmin (normalize=True) in one pass: returns (cost [B, Do, Ho, Wo],
    disp [B, 1, Ho, Wo]) with disp == soft_argmin(cost, disp_values, alpha, True) bit for bit."""
    sh = _lib.shim()
    if sh is not None:
        y, disp = sh.trilinear_ac_soft_argmin(x if x.is_contiguous() else x.contiguous(), int(out_size[0]), int(out_size[1]), int(out_size[2]),
                                              float(alpha), [float(v) for v in disp_values])
        return y, disp
    lib = _lib.load()
    x = _f32c(x, "x")
    B, Di, Hi, Wi = x.shape
    Do, Ho, Wo = out_size
    y = torch.empty((B, Do, Ho, Wo), dtype=torch.float32, device=x.device)
    disp = torch.empty((B, 1, Ho, Wo), dtype=torch.float32, device=x.device)
    check(lib.dmb_trilinear_ac_soft_argmin_f32(dev_ptr(x), dev_ptr(y), dev_ptr(disp), B, Di, Hi, Wi, Do, Ho, Wo,
                                               float(alpha), host_floats(disp_values), stream_ptr(x.device)),
          "dmb_trilinear_ac_soft_argmin_f32")
    return y, disp


def deconv3d_k8s4_c1_soft_argmin(x, w, disp_values=None, alpha=1.0):
    """AcfNet's learned 4x up-sampling (ConvTranspose3d(1, 1, 8, 4, 2)) of x [B, D, H, W] AND, when ``disp_values`` is given,
    the soft-argmin (normalize=True) of the volume it writes, in one pass: returns (cost [B, 4D, 4H, 4W], disp or None)."""
    lib = _lib.load()
    x, w = _f32c(x, "x"), _f32c(w, "weight")
    if x.dim() != 4:
        raise _lib.DmbLibraryError("deconv3d_k8s4_c1_soft_argmin: x must be [B, D, H, W], got %s" % (tuple(x.shape),))
    B, D, H, W = x.shape
    if w.numel() != 512:   # the kernel reads 8 x 8 x 8 weights
        raise _lib.DmbLibraryError("deconv3d_k8s4_c1_soft_argmin: weight must hold 8x8x8 values, got %s" % (tuple(w.shape),))
    if disp_values is not None and len(disp_values) != 4 * D:   # ... and 4 D host floats
        raise _lib.DmbLibraryError("deconv3d_k8s4_c1_soft_argmin: %d disparity samples for %d output planes" % (len(disp_values), 4 * D))
    y = torch.empty((B, 4 * D, 4 * H, 4 * W), dtype=torch.float32, device=x.device)
    disp = torch.empty((B, 1, 4 * H, 4 * W), dtype=torch.float32, device=x.device) if disp_values is not None else None
    check(lib.dmb_deconv3d_k8s4_c1_soft_argmin_f32(dev_ptr(x), dev_ptr(w), dev_ptr(y), dev_ptr(disp, allow_none=True), B, D, H, W,
                                                   float(alpha), host_floats(disp_values) if disp_values is not None else None,
                                                   stream_ptr(x.device)), "dmb_deconv3d_k8s4_c1_soft_argmin_f32")
    return y, disp


class RegressionHint:
    """Side channel from a cost producer to the soft-argmin predictors: a disparity map that the producing kernel
    already regressed from exactly this cost tensor (same sample values, alpha, normalize=True).  The predictor uses it
    only if every parameter matches and the tensor has not been modified since (``_version``)."""

    __slots__ = ("values", "alpha", "version", "disp")

    def __init__(self, values, alpha, version, disp):
        self.values, self.alpha, self.version, self.disp = tuple(values), float(alpha), version, disp

    @staticmethod
    def attach(cost, values, alpha, disp):
        cost._dmb_regression_hint = RegressionHint(values, alpha, cost._version, disp)
        return cost

    @staticmethod
    def lookup(cost, values, alpha, normalize):
        hint = getattr(cost, "_dmb_regression_hint", None)
        if hint is None or not normalize or hint.version != cost._version or hint.alpha != float(alpha) \
                or hint.values != tuple(values) or hint.disp.shape[0] != cost.shape[0]:
            return None
        return hint.disp


# ---------------------------------------------------------------------------------------------- conf head / metrics
def pack_conf_head_weights(w1):
    """Conv2d weight [Cm, D, 3, 3] -> fragment stream."""
    lib = _lib.load()
    w1 = _f32c(w1, "weight")
    Cm, D = w1.shape[0], w1.shape[1]
    wp = torch.empty((lib.dmb_conf_head_packed_floats(Cm, D),), dtype=torch.float32, device=w1.device)
    check(lib.dmb_conf_head_pack_weights_f32(dev_ptr(w1), dev_ptr(wp), Cm, D, stream_ptr(w1.device)),
          "dmb_conf_head_pack_weights_f32")
    return wp


def conf_head(cost, w1pack, scale, shift, w2):
    lib = _lib.load()
    cost = _f32c(cost, "cost")
    B, D, H, W = cost.shape
    Cm = scale.numel()
    conf = torch.empty((B, 1, H, W), dtype=torch.float32, device=cost.device)
    check(lib.dmb_conf_head_f32(dev_ptr(cost), dev_ptr(w1pack), dev_ptr(scale), dev_ptr(shift), dev_ptr(w2),
                                dev_ptr(conf), B, D, Cm, H, W, stream_ptr(cost.device)), "dmb_conf_head_f32")
    return conf


class UpsampleSource:
    """Side channel from AcfNet's learned up-sampling to the confidence head: the quarter-resolution volume and the
    ConvTranspose3d(1, 1, 8, 4, 2) weight a full-resolution cost tensor was produced from.  Used only while every tensor is
    unmodified (``_version``)."""

    __slots__ = ("c", "w8", "versions")

    def __init__(self, c, w8, cost):
        self.c, self.w8 = c, w8
        self.versions = (c._version, w8._version, w8.data_ptr(), cost._version)

    @staticmethod
    def attach(cost, c, w8):
        cost._dmb_k8s4_source = UpsampleSource(c, w8, cost)
        return cost

    @staticmethod
    def lookup(cost):
        src = getattr(cost, "_dmb_k8s4_source", None)
        if src is None or src.versions != (src.c._version, src.w8._version, src.w8.data_ptr(), cost._version):
            return None
        B, D, H, W = cost.shape
        if tuple(src.c.shape) != (B, D // 4, H // 4, W // 4) or D % 4 or H % 4 or W % 4:
            return None
        return src


_conf_composite = True


def set_conf_head_composite(flag):
    """False forces the confidence head's 3x3 convolution on the up-sampled volume itself everywhere."""
    global _conf_composite
    _conf_composite = bool(flag)


def conf_head_k8s4_weights(w1, w8):
    """Composed weights of (3x3 head convolution) o (k8 s4 p2 transposed convolution), see csrc/confhead.hip:
    K[(by, bx, m), z, ty, tx] = sum_{dy, dx, kz} w1[m, 4z - 2 + kz, dy, dx] * w8[kz, by + dy + 5 - 4 ty, bx + dx + 5 - 4 tx]
    (indices outside 0..7 / 0..D-1 contribute nothing), summed in FP64 and rounded once -> [16 * M, D / 4, 3, 3] FP32,
    output channel = (by * 4 + bx) * M + m.  Pure tensor algebra (runs on any device; pinned on CPU by
    tests/test_host_logic.py against the plain loop nest)."""
    M = w1.shape[0]
    dev = w1.device
    w1d = torch.nn.functional.pad(w1.detach().double(), (0, 0, 0, 0, 2, 2))        # D axis: index 4z + kz <-> D = 4z - 2 + kz
    w1g = w1d.unfold(1, 8, 4)                                                        # [M, Dq, 3(dy), 3(dx), 8(kz)]
    sel = torch.zeros((4, 3, 3, 8), dtype=torch.float64, device=dev)               # [phase, t, d, k]: k == phase + d + 5 - 4 t
    for b in range(4):
        for t in range(3):
            for d in range(3):
                k = b + d + 5 - 4 * t
                if 0 <= k <= 7:
                    sel[b, t, d, k] = 1.0
    K = torch.einsum("mzdxk,kpq,btdp,euxq->bemztu", w1g, w8.detach().double().view(8, 8, 8), sel, sel)
    return K.reshape(16 * M, w1.shape[1] // 4, 3, 3).float().contiguous()


def conf_head_k8s4_pack(w1, w8, scale, shift):
    """conf_head_k8s4_weights packed for dmb_conv2d_f32 in launches of 128 output channels (two phases x M = 64), with the
    folded BatchNorm affine tiled per launch, plus w1 transposed for the pixel-ring kernel."""
    M = w1.shape[0]
    K = conf_head_k8s4_weights(w1, w8)
    per = 128 // M                                                                   # phases per launch
    packs = [pack_conv2d_weights(K[i * 128:(i + 1) * 128].contiguous()) for i in range(16 * M // 128)]
    return {"packs": packs, "scale": scale.repeat(per).contiguous(), "shift": shift.repeat(per).contiguous(), "M": M,
            "w1t": w1.detach().float().permute(1, 2, 3, 0).contiguous()}


def conf_head_composite_applicable(cost, M):
    return _conf_composite and M == 64 and UpsampleSource.lookup(cost) is not None and cost.shape[3] % 16 == 0


_conf_dot_epilogue = True   # development switch: False = hidden tensor + dmb_conf_gather_f32


def set_conf_dot_epilogue(flag):
    global _conf_dot_epilogue
    _conf_dot_epilogue = bool(flag)


def conf_head_from_source(cost, comp, scale, shift, w2):
    """The confidence map of ``cost`` (which carries an UpsampleSource) through the composed quarter-resolution form."""
    lib = _lib.load()
    src = UpsampleSource.lookup(cost)
    c = _f32c(src.c, "quarter-resolution cost")
    B, Dq, Hq, Wq = c.shape
    M = comp["M"]
    conf = torch.empty((B, 1, 4 * Hq, 4 * Wq), dtype=torch.float32, device=c.device)
    if M == 64 and len(comp["packs"]) == 8 and _conf_dot_epilogue:
        # one launch over (spatial tile, weight set) work items, two phases per set; the 64-vector of every pixel is reduced against w2 in the convolution's epilogue, so the
        # [B, 1024, Hq, Wq] hidden tensor (0.5 GB at the BASELINE size) is never written
        w2c = _f32c(w2.reshape(-1), "w2")
        if "pack_all" not in comp:
            comp["pack_all"] = torch.cat([wp.reshape(-1) for wp in comp["packs"]]).contiguous()   # the 8 weight sets back to back
        check(lib.dmb_conf_phase_conv2d_f32(dev_ptr(c), dev_ptr(comp["pack_all"]), dev_ptr(comp["scale"]), dev_ptr(comp["shift"]),
                                            dev_ptr(w2c), dev_ptr(conf), B, Dq, Hq, Wq, 8, stream_ptr(c.device)), "dmb_conf_phase_conv2d_f32")
    else:
        hq = torch.empty((B, 16 * M, Hq, Wq), dtype=torch.float32, device=c.device)
        for i, wp in enumerate(comp["packs"]):
            conv2d(c, wp, 128, 3, scale=comp["scale"], shift=comp["shift"], relu=True, out=hq, out_ch_offset=128 * i)
        check(lib.dmb_conf_gather_f32(dev_ptr(hq), dev_ptr(w2), dev_ptr(conf), B, M, Hq, Wq, stream_ptr(c.device)), "dmb_conf_gather_f32")
    check(lib.dmb_conf_ring_f32(dev_ptr(cost), dev_ptr(comp["w1t"]), dev_ptr(scale), dev_ptr(shift), dev_ptr(w2), dev_ptr(conf),
                                B, 4 * Dq, M, 4 * Hq, 4 * Wq, stream_ptr(c.device)), "dmb_conf_ring_f32")
    return conf


EPE_WORKSPACE_DOUBLES_PER_IMAGE = 64 * 6   # include/dmb_hip.h: 64 slices x 6 sums per image (and estimate)


def epe_accumulate(est, gt, acc, original_size, lower_bound, upper_bound):
    """acc: float64[6] device tensor updated in place; est/gt: [B, 1, Hp, Wp]."""
    lib = _lib.load()
    est, gt = _f32c(est, "est_disp"), _f32c(gt, "gt_disp")
    B = est.shape[0]
    Hp, Wp = est.shape[-2:]
    H0, W0 = original_size
    if acc.dtype != torch.float64 or acc.numel() != 6:
        raise _lib.DmbLibraryError("acc must be a float64[6] tensor")
    _same_shape(est, gt, "epe_accumulate: estimate and ground truth")
    if est.numel() != B * Hp * Wp or not (0 < int(H0) <= Hp and 0 < int(W0) <= Wp):
        raise _lib.DmbLibraryError("epe_accumulate: maps must be [B, 1, Hp, Wp] with the original size inside, got %s / %s"
                                   % (tuple(est.shape), (H0, W0)))
    ws = torch.empty((B, EPE_WORKSPACE_DOUBLES_PER_IMAGE), dtype=torch.float64, device=est.device)
    check(lib.dmb_epe_accum_f64(dev_ptr(est), dev_ptr(gt), dev_ptr(acc), dev_ptr(ws), B, Hp, Wp, int(H0), int(W0),
                                float(lower_bound), float(upper_bound), stream_ptr(est.device)), "dmb_epe_accum_f64")
    return acc


def epe_accumulate_multi(ests, gt, acc, original_size, lower_bound, upper_bound):
    """``ests``: 1 .. 4 maps [B, 1, Hp, Wp] evaluated against the same ``gt`` in one pass; acc: float64[len(ests), 6] (contiguous
    rows of a device tensor), row i updated from ests[i] exactly as ``epe_accumulate`` would."""
    import ctypes
    lib = _lib.load()
    n = len(ests)
    ests = [_f32c(e, "est_disp") for e in ests]
    gt = _f32c(gt, "gt_disp")
    B = gt.shape[0]
    Hp, Wp = gt.shape[-2:]
    H0, W0 = original_size
    if n < 1 or n > 4 or acc.dtype != torch.float64 or tuple(acc.shape) != (n, 6) or not acc.is_contiguous():
        raise _lib.DmbLibraryError("epe_accumulate_multi: 1 .. 4 estimates and a contiguous float64[n, 6] accumulator")
    for e in ests:
        _same_shape(e, gt, "epe_accumulate_multi: estimate and ground truth")
    if gt.numel() != B * Hp * Wp or not (0 < int(H0) <= Hp and 0 < int(W0) <= Wp):
        raise _lib.DmbLibraryError("epe_accumulate_multi: maps must be [B, 1, Hp, Wp] with the original size inside, got %s / %s"
                                   % (tuple(gt.shape), (H0, W0)))
    ws = torch.empty((n, B, EPE_WORKSPACE_DOUBLES_PER_IMAGE), dtype=torch.float64, device=gt.device)
    PA = ctypes.c_void_p * n
    check(lib.dmb_epe_accum_multi_f64(n, PA(*[dev_ptr(e).value for e in ests]), dev_ptr(gt), dev_ptr(acc), dev_ptr(ws), B, Hp, Wp,
                                      int(H0), int(W0), float(lower_bound), float(upper_bound), stream_ptr(gt.device)),
          "dmb_epe_accum_multi_f64")
    return acc


# ---------------------------------------------------------------------------------------------- 2-D backbone ops
def _window_ptr(t, ch_offset):
    """Pointer to channel ``ch_offset`` of batch item 0 of a contiguous [B, C, H, W] tensor."""
    import ctypes
    dev_ptr(t)  # validates device / dtype / contiguity
    return ctypes.c_void_p(t.data_ptr() + 4 * ch_offset * t.shape[2] * t.shape[3])


def pack_conv2d_weights(w):
    """nn.Conv2d weight [Co, Ci, k, k] (k in {1, 3, 5}) -> MFMA A-fragment stream."""
    lib = _lib.load()
    w = _f32c(w, "weight")
    Co, Ci, k = w.shape[0], w.shape[1], w.shape[2]
    wp = torch.empty((lib.dmb_conv2d_packed_floats(Co, Ci, k),), dtype=torch.float32, device=w.device)
    check(lib.dmb_conv2d_pack_weights_f32(dev_ptr(w), dev_ptr(wp), Co, Ci, k, stream_ptr(w.device)),
          "dmb_conv2d_pack_weights_f32")
    return wp


def conv2d(x, wpack, Co, ksize, stride=1, dilation=1, scale=None, shift=None, residual=None, relu=False,
           in_window=None, out=None, out_ch_offset=0, res_ch_offset=0):
    """x: [B, Cx, H, W]; ``in_window=(offset, Ci)`` reads channels [offset, offset + Ci) of it (default: all).
    ``out``: optional pre-allocated [B, Ctot, Ho, Wo] tensor written at channel ``out_ch_offset``."""
    sh = _lib.shim()
    if sh is not None:
        coff, Ci = in_window if in_window is not None else (0, x.shape[1])
        return sh.conv2d(x if x.is_contiguous() else x.contiguous(), coff, Ci, wpack, Co, ksize, stride, dilation, scale, shift, residual,
                         res_ch_offset, int(bool(relu)) | _conv_flags(), out, out_ch_offset)
    lib = _lib.load()
    x = _f32c(x, "x")
    B, Cx, H, W = x.shape
    coff, Ci = in_window if in_window is not None else (0, Cx)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if out is None:
        out = torch.empty((B, Co, Ho, Wo), dtype=torch.float32, device=x.device)
        out_ch_offset = 0
    if tuple(out.shape[2:]) != (Ho, Wo) or out.shape[0] != B or out_ch_offset + Co > out.shape[1]:
        raise _lib.DmbLibraryError("conv2d: output tensor %s does not fit" % (tuple(out.shape),))
    if residual is not None and (tuple(residual.shape[2:]) != (Ho, Wo) or residual.shape[1] < res_ch_offset + Co):
        raise _lib.DmbLibraryError("conv2d: residual shape %s" % (tuple(residual.shape),))
    check(lib.dmb_conv2d_f32(_window_ptr(x, coff), dev_ptr(wpack), dev_ptr(scale, allow_none=True),
                             dev_ptr(shift, allow_none=True),
                             _window_ptr(residual, res_ch_offset) if residual is not None else None,
                             _window_ptr(out, out_ch_offset), B, Ci, Co, H, W, ksize, stride, dilation, int(bool(relu)) | _conv_flags(),
                             Cx, out.shape[1], residual.shape[1] if residual is not None else 0, stream_ptr(x.device)),
          "dmb_conv2d_f32")
    return out


def conv2d_k3_multi(jobs, Co):
    """``jobs``: list of (x [B, Ci, H, W_q], wpack, out [B, Ctot_q, H, W_q], out_ch_offset): independent 3x3 stride-1 convolutions
    of one layer shape in ONE launch (csrc/conv2d.hip, MULTI); no affine / residual / ReLU."""
    import ctypes
    lib = _lib.load()
    n = len(jobs)
    if n < 1 or n > 6:
        raise _lib.DmbLibraryError("conv2d_k3_multi: 1 .. 6 jobs per launch, got %d" % n)
    x0 = _f32c(jobs[0][0], "x")
    B, Ci, H = x0.shape[0], x0.shape[1], x0.shape[2]
    xs, ws, ys, Ws, cts = [], [], [], [], []
    keep = []   # converted inputs stay alive until the launch is enqueued (a freed temporary's block could be handed to the next job)
    for x, wp, out, off in jobs:
        x = _f32c(x, "x")
        keep.append(x)
        if tuple(x.shape[:3]) != (B, Ci, H) or tuple(out.shape[2:]) != tuple(x.shape[2:]) or out.shape[0] != B or off + Co > out.shape[1]:
            raise _lib.DmbLibraryError("conv2d_k3_multi: job shapes %s -> %s do not fit" % (tuple(x.shape), tuple(out.shape)))
        if out.dtype != torch.float32 or not out.is_contiguous() or out.device != x0.device or x.device != x0.device:
            raise _lib.DmbLibraryError("conv2d_k3_multi: outputs must be contiguous float32 tensors on %s" % (x0.device,))
        if wp.dtype != torch.float32 or not wp.is_contiguous() or wp.device != x0.device:
            raise _lib.DmbLibraryError("conv2d_k3_multi: packed weights must be contiguous float32 tensors on %s" % (x0.device,))
        xs.append(dev_ptr(x).value)
        ws.append(dev_ptr(wp).value)
        ys.append(_window_ptr(out, off).value)
        Ws.append(x.shape[3])
        cts.append(out.shape[1])
    PA, IA = ctypes.c_void_p * n, ctypes.c_int * n
    check(lib.dmb_conv2d_k3_multi_f32(n, PA(*xs), PA(*ws), PA(*ys), IA(*Ws), IA(*cts), B, Ci, Co, H, stream_ptr(x0.device)),
          "dmb_conv2d_k3_multi_f32")


def avgpool2d(x, k, in_window=None):
    lib = _lib.load()
    x = _f32c(x, "x")
    B, Cx, H, W = x.shape
    coff, C = in_window if in_window is not None else (0, Cx)
    y = torch.empty((B, C, H // k, W // k), dtype=torch.float32, device=x.device)
    check(lib.dmb_avgpool2d_f32(dev_ptr(x), dev_ptr(y), B, C, H, W, int(k), Cx, coff, stream_ptr(x.device)),
          "dmb_avgpool2d_f32")
    return y


def bilinear_ac(x, out_hw, out=None, out_ch_offset=0):
    lib = _lib.load()
    x = _f32c(x, "x")
    B, C, Hi, Wi = x.shape
    Ho, Wo = out_hw
    if out is None:
        out = torch.empty((B, C, Ho, Wo), dtype=torch.float32, device=x.device)
        out_ch_offset = 0
    check(lib.dmb_bilinear_ac_f32(dev_ptr(x), dev_ptr(out), B, C, Hi, Wi, Ho, Wo, out.shape[1], out_ch_offset,
                                  stream_ptr(x.device)), "dmb_bilinear_ac_f32")
    return out


def bilinear_scale(x, out_hw, mult=1.0, out=None, out_ch_offset=0):
    """F.interpolate(x, out_hw, mode='bilinear', align_corners=False) * mult."""
    lib = _lib.load()
    x = _f32c(x, "x")
    B, C, Hi, Wi = x.shape
    Ho, Wo = out_hw
    if out is None:
        out = torch.empty((B, C, Ho, Wo), dtype=torch.float32, device=x.device)
        out_ch_offset = 0
    check(lib.dmb_bilinear_scale_f32(dev_ptr(x), dev_ptr(out), B, C, Hi, Wi, Ho, Wo, float(mult), out.shape[1],
                                     out_ch_offset, stream_ptr(x.device)), "dmb_bilinear_scale_f32")
    return out


# ---------------------------------------------------------------------------------------------- data-side conventions
IMAGENET_MEAN = (123.675, 116.28, 103.53)     # dmb/apis/inference.py:120-121, configs/PSMNet/scene_flow.py:94 (0..255 scale)
IMAGENET_STD = (58.395, 57.12, 57.375)


def stereo_pad_normalize(src, size=None, mean=None, std=None, window=None, channels=None, out=None):
    """CenterCrop window -> StereoPad (zeros on the top and on the right) -> Normalize, one launch (csrc/preprocess.hip;
    dmb/data/transforms/stereo_trans.py:20-44,78-119 in the order of dmb/data/datasets/stereo/builder.py:22-28).
    ``src``: float32 [B, Cs, sh, sw], or the decoder's uint8 [B, sh, sw, Cs] (the first ``channels`` are taken, default
    min(Cs, 3)); ``window`` = (y0, x0, h, w) of the source kept (default: all of it); ``size`` = (th, tw) >= window (default:
    the window's); ``mean`` / ``std``: per-channel sequences (both or neither).  Returns float32 [B, C, th, tw]."""
    lib = _lib.load()
    if not isinstance(src, torch.Tensor) or not src.is_cuda:
        raise _lib.DmbLibraryError("stereo_pad_normalize: the source must be a GPU tensor (no CPU fallback), got %s"
                                   % (getattr(src, "device", type(src)),))
    if src.dim() != 4 or src.dtype not in (torch.float32, torch.uint8):
        raise _lib.DmbLibraryError("stereo_pad_normalize: float32 [B, C, H, W] or uint8 [B, H, W, C] expected, got %s %s"
                                   % (src.dtype, tuple(src.shape)))
    src = src.contiguous()
    u8 = src.dtype == torch.uint8
    if u8:
        B, sh, sw, Cs = src.shape
        C = min(Cs, 3) if channels is None else int(channels)
    else:
        B, Cs, sh, sw = src.shape
        C = Cs if channels is None else int(channels)
    y0, x0, h, w = (0, 0, sh, sw) if window is None else [int(v) for v in window]
    th, tw = (h, w) if size is None else [int(v) for v in size]
    if (mean is None) != (std is None) or (mean is not None and (len(mean) != C or len(std) != C)):
        raise _lib.DmbLibraryError("stereo_pad_normalize: mean and std must both hold %d values" % C)
    y = _out_tensor(out, (B, C, th, tw), src.device, "stereo_pad_normalize")
    stream_ptr(src.device)     # (validates the device)
    fn = lib.dmb_stereo_pad_normalize_u8 if u8 else lib.dmb_stereo_pad_normalize_f32
    check(fn(ctypes.c_void_p(src.data_ptr()), dev_ptr(y), B, C, Cs, sh, sw, y0, x0, h, w, th, tw,
             host_floats(mean) if mean is not None else None, host_floats(std) if std is not None else None,
             stream_ptr(src.device)), "dmb_stereo_pad_normalize")
    return y


# ---------------------------------------------------------------------------------------------- training-side losses
def _loss_workspace(n, device):
    lib = _lib.load()
    return torch.empty((lib.dmb_loss_workspace_doubles(int(n)),), dtype=torch.float64, device=device)


def stereo_focal_loss_fwd(cost, gt, variance, disp_values, lower, upper, start_disp, end_disp, focal_coefficient):
    """Returns (loss_out [2] = (loss, divisor), stats [B, H, W, 2]); ``variance``: float or [B, 1, H, W] tensor."""
    lib = _lib.load()
    cost, gt = _f32c(cost, "cost"), _f32c(gt, "gt")
    B, D, H, W = cost.shape
    vmap = _f32c(variance, "variance") if torch.is_tensor(variance) else None
    if gt.numel() != B * H * W or len(disp_values) != D or (vmap is not None and vmap.numel() != B * H * W):
        raise _lib.DmbLibraryError("stereo_focal_loss_fwd: cost %s needs gt / variance maps of %d elements and %d disparity values, got %s / %s / %d"
                                   % (tuple(cost.shape), B * H * W, D, tuple(gt.shape),
                                      tuple(vmap.shape) if vmap is not None else None, len(disp_values)))
    stats = torch.empty((B, H, W, 2), dtype=torch.float32, device=cost.device)
    out = torch.empty((2,), dtype=torch.float32, device=cost.device)
    check(lib.dmb_stereo_focal_loss_fwd_f32(dev_ptr(cost), dev_ptr(gt), dev_ptr(vmap, allow_none=True),
                                            1.0 if vmap is not None else float(variance), host_floats(disp_values),
                                            dev_ptr(stats), dev_ptr(_loss_workspace(B * H * W, cost.device)), dev_ptr(out),
                                            B, D, H, W, float(lower), float(upper), float(start_disp), float(end_disp),
                                            float(focal_coefficient), stream_ptr(cost.device)),
          "dmb_stereo_focal_loss_fwd_f32")
    return out, stats


def stereo_focal_loss_bwd(cost, gt, variance, disp_values, stats, loss_out, grad_out, lower, upper, start_disp, end_disp,
                          focal_coefficient, want_grad_variance):
    lib = _lib.load()
    cost, gt = _f32c(cost, "cost"), _f32c(gt, "gt")
    B, D, H, W = cost.shape
    vmap = _f32c(variance, "variance") if torch.is_tensor(variance) else None
    if gt.numel() != B * H * W or len(disp_values) != D or (vmap is not None and vmap.numel() != B * H * W):
        raise _lib.DmbLibraryError("stereo_focal_loss_bwd: cost %s needs gt / variance maps of %d elements and %d disparity values, got %s / %s / %d"
                                   % (tuple(cost.shape), B * H * W, D, tuple(gt.shape),
                                      tuple(vmap.shape) if vmap is not None else None, len(disp_values)))
    gcost = torch.empty_like(cost)
    gvar = torch.empty((B, 1, H, W), dtype=torch.float32, device=cost.device) if (want_grad_variance and vmap is not None) else None
    go = _f32c(grad_out.reshape(1), "grad_out")
    check(lib.dmb_stereo_focal_loss_bwd_f32(dev_ptr(cost), dev_ptr(gt), dev_ptr(vmap, allow_none=True),
                                            1.0 if vmap is not None else float(variance), host_floats(disp_values),
                                            dev_ptr(stats), dev_ptr(loss_out), dev_ptr(go), 1.0, dev_ptr(gcost),
                                            dev_ptr(gvar, allow_none=True), B, D, H, W, float(lower), float(upper),
                                            float(start_disp), float(end_disp), float(focal_coefficient),
                                            stream_ptr(cost.device)), "dmb_stereo_focal_loss_bwd_f32")
    return gcost, gvar


def map_loss_fwd(x, gt, lower, upper, mode):
    lib = _lib.load()
    x, gt = _f32c(x, "x"), _f32c(gt, "gt")
    if gt.numel() != x.numel():
        raise _lib.DmbLibraryError("map_loss_fwd: map %s and ground truth %s differ in size" % (tuple(x.shape), tuple(gt.shape)))
    out = torch.empty((2,), dtype=torch.float32, device=x.device)
    check(lib.dmb_map_loss_fwd_f32(dev_ptr(x), dev_ptr(gt), dev_ptr(_loss_workspace(x.numel(), x.device)), dev_ptr(out),
                                   x.numel(), float(lower), float(upper), int(mode), stream_ptr(x.device)),
          "dmb_map_loss_fwd_f32")
    return out


def map_loss_bwd(x, gt, loss_out, grad_out, lower, upper, mode):
    lib = _lib.load()
    x, gt = _f32c(x, "x"), _f32c(gt, "gt")
    if gt.numel() != x.numel():
        raise _lib.DmbLibraryError("map_loss_bwd: map %s and ground truth %s differ in size" % (tuple(x.shape), tuple(gt.shape)))
    gx = torch.empty_like(x)
    go = _f32c(grad_out.reshape(1), "grad_out")
    check(lib.dmb_map_loss_bwd_f32(dev_ptr(x), dev_ptr(gt), dev_ptr(loss_out), dev_ptr(go), 1.0, dev_ptr(gx), x.numel(),
                                   float(lower), float(upper), int(mode), stream_ptr(x.device)), "dmb_map_loss_bwd_f32")
    return gx


# ---------------------------------------------------------------------------------------------- spatial propagation (dmb.ops.spn)
def spn_gaterecurrent2d(X, G1, G2, G3, horizontal, reverse):
    """dmb/ops/spn/functions/gaterecurrent2dnoind.py:10-24 (forward): the whole scan in one launch (csrc/spn.hip)."""
    lib = _lib.load()
    X = _f32c(X, "X")
    ts = [_f32c(g, "gate") for g in (G1, G2, G3)]
    for g in ts:
        _same_shape(X, g, "spn")
    if X.dim() != 4:
        raise _lib.DmbLibraryError("spn: X must be [N, C, H, W], got %s" % (tuple(X.shape),))
    N, C, H, W = X.shape
    out = torch.empty_like(X)
    check(lib.dmb_spn_gaterecurrent2d_f32(dev_ptr(X), dev_ptr(ts[0]), dev_ptr(ts[1]), dev_ptr(ts[2]), dev_ptr(out), N, C, H, W,
                                          1 if horizontal else 0, 1 if reverse else 0, stream_ptr(X.device)), "dmb_spn_gaterecurrent2d_f32")
    return out


def spn_gaterecurrent2d_bwd(X, G1, G2, G3, H_fwd, grad_out, horizontal, reverse):
    """gaterecurrent2dnoind.py:26-44 (backward): (dX, dG1, dG2, dG3)."""
    lib = _lib.load()
    ops_in = [_f32c(t, "spn operand") for t in (X, G1, G2, G3, H_fwd, grad_out)]
    for t in ops_in[1:]:
        _same_shape(ops_in[0], t, "spn_bwd")
    N, C, H, W = ops_in[0].shape
    outs = [torch.empty_like(ops_in[0]) for _ in range(4)]
    check(lib.dmb_spn_gaterecurrent2d_bwd_f32(*[dev_ptr(t) for t in ops_in], *[dev_ptr(t) for t in outs], N, C, H, W,
                                              1 if horizontal else 0, 1 if reverse else 0, stream_ptr(ops_in[0].device)),
          "dmb_spn_gaterecurrent2d_bwd_f32")
    return tuple(outs)


from .spn import GateRecurrent2dnoind  # noqa: E402,F401  (``dmb.ops.GateRecurrent2dnoind``: dmb/ops/__init__.py:1)
