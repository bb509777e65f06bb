"""Autograd of the cost-path units on the HIP kernels (second part).

What torch.autograd does for the reference's ``nn.Sequential(Conv3d | ConvTranspose3d, BatchNorm3d[, ReLU])`` units
(dmb/modeling/stereo/layers/basic_layers.py:68-100,160-177), its skip adds (cost_processors/utils/hourglass.py:62-86),
the cost-volume builders (cost_processors/utils/cat_fms.py:7-48, dif_fms.py:7-46) and the up-sampling + soft-argmin
tail (aggregators/PSMNet.py:74-93, disp_predictors/faster_soft_argmin.py:51-75), expressed as ``autograd.Function``s
whose forward and backward are HIP kernel launches:

  forward   raw = conv(x, w) (+ bias)              the inference kernel without its epilogue
            scale/shift from batch statistics (training) or the running buffers (eval)      bn_train_stats
            y = act(raw*scale + shift (+ skip))                                             bn_act
  backward  dc, dgamma, dbeta, dskip <- dy                                                  bn_act_bwd
            dw = wgrad(x, dc)                                                               conv3d_k3[s2]_wgrad
            dx = the forward kernel of the adjoint convolution on re-packed weights         conv3d_k3 / deconv3d_k3s2

There is no fallback to torch's convolution backward: without the library these raise like every other op.
"""
import contextlib
import threading
import weakref

import torch

from .... import ops

_RELU = {0: False, 1: True, 2: "pre"}
_CONST = {}

# ---------------------------------------------------------------------------------------------- gradient carry (round 6)
# A tensor with several consumers (the skip operands of hourglass.py:62-86 and PSMNet.py:58-72: cost0 has four, pre1 three)
# gets its gradient as a SUM over them, and torch.autograd forms that sum with one addition launch per extra consumer -- three
# passes over a 200 MB tensor each, 18 of them per PSMNet step (0.75 ms of 31).  Inside a ``carry_scope()`` every unit hands
# the tensors it consumes on as identity outputs ("aliases": the same storage, a new autograd edge) and the NEXT consumer of the
# tensor is quietly given the alias instead.  The consumers then form a chain, each one's backward receives what the later
# consumers have already collected, and adds its own share inside a kernel that runs anyway: the data-gradient convolution
# takes it as its skip operand (one extra read), the BatchNorm backward as ``dres_acc``.  Same gradients up to the order of
# the FP32 additions; ``set_gradient_carry(False)`` switches it off (tests/test_train_gpu.py compares the two).
_carry_enabled = True
_tls = threading.local()      # the open scope's registry belongs to the thread that runs the forward pass


def set_gradient_carry(flag):
    global _carry_enabled
    _carry_enabled = bool(flag)


def _registry():
    return getattr(_tls, "registry", None)


@contextlib.contextmanager
def carry_scope():
    """One forward pass of a model / aggregator.  The registry maps id(tensor) -> (tensor, its latest alias) and lives exactly as
    long as the scope (a tensor object the caller re-uses across steps must never meet an alias of an earlier graph); nested
    scopes share the outermost registry; one registry per thread."""
    if _registry() is not None:
        yield
        return
    _tls.registry = {}
    try:
        yield
    finally:
        _tls.registry.clear()
        _tls.registry = None


def _latest(reg, t):
    while True:
        e = reg.get(id(t))
        if e is None:
            return t
        t = e[1]


def _carry_plan(x, skip):
    """(registry or None, x, skip, carry_x, carry_skip) for a unit about to consume x (and skip)."""
    reg = _registry()
    if reg is None or not _carry_enabled or not torch.is_grad_enabled():
        return None, x, skip, False, False
    x = _latest(reg, x)
    if skip is not None:
        skip = _latest(reg, skip)
    cx = x.requires_grad and x.is_contiguous() and x.dtype == torch.float32
    cs = skip is not None and skip is not x and skip.requires_grad and skip.is_contiguous() and skip.dtype == torch.float32
    return reg, x, skip, cx, cs


def _const(value, C, device):
    """Cached [C] tensor of ones / zeros (identity scale / shift of units without BatchNorm): no fill launch per call."""
    key = (float(value), int(C), str(device))
    t = _CONST.get(key)
    if t is None:
        t = _CONST[key] = torch.full((C,), float(value), dtype=torch.float32, device=device)
    return t


def _relu_code(relu):
    return 2 if relu == "pre" else (1 if relu else 0)


def _mask_mode(code, has_skip):
    """Which tensor the backward takes the ReLU mask from.  Without a skip operand ``y = max(fma(raw, scale, shift), 0)``, so
    ``y > 0`` IS ``fma(raw, scale, shift) > 0`` bit for bit: the backward passes re-create the mask from ``raw`` (which they read
    anyway) and the unit's output is neither read again nor kept alive for it -- one tensor less in each of the two passes for
    16 of PSMNet's 25 units."""
    return _RELU[2 if code == 1 and not has_skip else code]


class _PackGroup:
    """Packed weights of every 3-D unit that has taken the training path on one device: the forward AND the data-gradient pack of
    all of them are re-made by ONE launch per forward pass (ops.run_pack_table) instead of one launch per unit and direction (52
    launches of 4.8 us in a PSMNet step).  The pack buffers and the device table persist; units are held weakly.

    When to re-pack.  A weight's ``_version`` moving is one trigger, but torch's FUSED optimizers (``Adam(fused=True)``) update
    parameters without moving it -- so the group also counts passes: a unit asking for its packs a second time within one
    generation means a new forward pass has begun (every unit runs once per pass), the generation advances and everything is
    re-packed, whatever the versions say.  One 19 us launch per pass either way."""

    def __init__(self, device):
        self.device, self.entries, self.table, self.njobs, self.gen = device, {}, None, 0, 0

    def packs(self, unit, w):
        e = self.entries.get(id(unit))
        if e is None or e["unit"]() is not unit or e["ptr"] != w.data_ptr() or e["shape"] != tuple(w.shape):
            jobs = ops.unit_pack_jobs(w, unit.transposed, unit.stride)
            e = {"unit": weakref.ref(unit), "ptr": w.data_ptr(), "shape": tuple(w.shape), "version": None, "seen": -1, "packed": -1,
                 "jobs": jobs,
                 "bufs": [torch.empty((ops.packed_floats(co, ci),), dtype=torch.float32, device=w.device) for co, ci, _ in jobs]}
            self.entries[id(unit)] = e
            self.table = None
        if e["seen"] == self.gen:
            self.gen += 1                # second request within a generation: a new forward pass
        e["seen"] = self.gen
        if e["packed"] != self.gen or e["version"] != w._version or self.table is None:
            self._repack(unit, w)
        return e["bufs"]

    def _repack(self, unit, w):
        if self.table is None:
            jobs, keep = [], {}
            for k, e in self.entries.items():
                u = e["unit"]()
                if u is None:
                    continue
                wt = w if u is unit else u[0].weight.detach()
                if wt.data_ptr() != e["ptr"] or tuple(wt.shape) != e["shape"] or not wt.is_contiguous() or wt.device != self.device:
                    continue            # replaced since (load_state_dict keeps pointers; .to() / re-assignment does not): re-registers itself
                keep[k] = e
                for (co, ci, mode), buf in zip(e["jobs"], e["bufs"]):
                    jobs.append((wt, buf, co, ci, mode))
            self.entries = keep
            self.table, self.njobs = ops.make_pack_table(jobs, self.device), len(jobs)
        ops.run_pack_table(self.table, self.njobs)
        dead = False
        for e in self.entries.values():
            u = e["unit"]()
            e["packed"] = self.gen
            e["version"] = None if u is None else u[0].weight._version
            dead = dead or u is None
        if dead:
            self.table = None           # a unit died: its job still ran (the buffers are ours), the next re-pack leaves it out


_pack_groups = {}
_pack_group_enabled = True


def set_pack_group(flag):
    """False: one pack launch per unit, direction and call, as in rounds 1-5 (A/B and tests; the 2-D units' phase packs too)."""
    global _pack_group_enabled
    _pack_group_enabled = bool(flag)


def _unit_packs(unit, w):
    """(forward pack, data-gradient pack) of a unit's weight; the data-gradient pack may be None (made on demand)."""
    if not _pack_group_enabled or torch.cuda.is_current_stream_capturing():
        fwd = ops.pack_deconv3d_weights(w) if unit.transposed else ops.pack_conv3d_weights(w)
        return fwd, None
    key = (w.device.type, w.device.index)
    g = _pack_groups.get(key)
    if g is None:
        g = _pack_groups[key] = _PackGroup(w.device)
    return tuple(g.packs(unit, w))


def _conv_raw(unit, x, wpack, bias):
    """The unit's convolution without BatchNorm / activation (+ bias through the kernel's shift operand)."""
    Co = unit.out_planes
    scale = shift = None
    if bias is not None:
        scale, shift = _const(1.0, bias.numel(), bias.device), bias.detach().contiguous()
    if unit.transposed:
        return ops.deconv3d_k3s2(x, wpack, Co, scale, shift, None, False)
    return ops.conv3d_k3(x, wpack, Co, scale, shift, None, unit.stride, False)


# Batch statistics out of the convolution's epilogue (conv3d_s1_kernel's STATS instantiation + ops.bn_train_act; VERDICT round 5,
# "not attempted" twice).  Built, tested -- and OFF by default, because it does not pay: at 4 x 256x512 the six units it covers lose
# their 33 us statistics pass each (-0.20 ms) but the convolution's epilogue grows by 16 us (+0.10 ms: the FP64 sums, the lane
# butterfly and a barrier sit in the one part of that kernel nothing overlaps) and the normalising pass, which now finishes 3072
# partials per channel instead of 128, by 8 us (+0.05 ms): 27.2 ms per step either way (docs/design/12-6).
_epilogue_stats = False


def set_epilogue_stats(flag):
    """True: 32-channel stride-1 units take the block sums of their batch statistics from the convolution's epilogue."""
    global _epilogue_stats
    _epilogue_stats = bool(flag)


def _unit_bn_forward(bn, raw, gamma, beta, skip, code, C, device, partials=None):
    """BatchNorm (+ skip, + ReLU) of a unit's raw convolution output -> (y, mean, invstd, scale, shift, batch_stats).  A
    batch-statistics BatchNorm takes the two-launch form (ops.bn_train_fwd: block sums, then one kernel that finishes the
    statistics, updates the running buffers and the batch counter and normalises); running statistics / no BatchNorm: bn_act."""
    if bn is not None and bn.training:
        if bn.momentum is None:   # nn.BatchNorm: cumulative moving average, factor 1 / (batches seen including this one)
            momentum = 1.0 / float(int(bn.num_batches_tracked) + 1) if bn.track_running_stats else 0.0
        else:
            momentum = bn.momentum
        nbt = bn.num_batches_tracked if bn.track_running_stats and bn.num_batches_tracked is not None else None
        if nbt is not None and nbt.device != raw.device:
            nbt += 1
            nbt = None
        args = (gamma.detach() if gamma is not None else None, beta.detach() if beta is not None else None,
                bn.running_mean if bn.track_running_stats else None, bn.running_var if bn.track_running_stats else None, nbt, momentum,
                bn.eps, skip, _RELU[code])
        if partials is not None:   # the convolution's epilogue already summed its output: no pass for the block sums
            y, mean, invstd, scale, shift = ops.bn_train_act(raw, partials, *args)
        else:
            y, mean, invstd, scale, shift = ops.bn_train_fwd(raw, *args)
        return y, mean, invstd, scale, shift, True
    mean, invstd, scale, shift, _ = _bn_forward(bn, False, raw, gamma, beta, C, device)
    y = raw if (bn is None and skip is None and code == 0) else ops.bn_act(raw, scale, shift, skip, _RELU[code])
    return y, mean, invstd, scale, shift, False


def _carried_outputs(ctx, y, x, skip, carry_x, carry_skip):
    ctx.carry = (bool(carry_x), bool(carry_skip))
    if not (carry_x or carry_skip):
        return y
    ctx.set_materialize_grads(False)   # an alias nobody consumed has no gradient: None, not a tensor of zeros
    return (y,) + ((x,) if carry_x else ()) + ((skip,) if carry_skip else ())


def _carried_grads(ctx, grads):
    """(dy, what x's later consumers collected, what skip's later consumers collected) from backward's arguments."""
    g = list(grads)
    dy = g.pop(0)
    dxa = g.pop(0) if ctx.carry[0] else None
    dsa = g.pop(0) if ctx.carry[1] else None
    return dy, dxa, dsa


class ConvUnitFn(torch.autograd.Function):
    """y = act(BN(conv(x)) (+ skip)) of one FusedConv3d unit; ``unit`` supplies the geometry and the BN buffers.  With
    ``carry_x`` / ``carry_skip`` the inputs are handed on as identity outputs (gradient carry, top of this file)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, skip, unit, relu, carry_x=False, carry_skip=False):
        x_in = x
        x = x.contiguous()
        w = weight.detach().contiguous()
        wp_fwd, wp_bwd = _unit_packs(unit, w)
        C = unit.out_planes
        bn = unit[1] if unit.has_bn else None
        partials = None
        if (_epilogue_stats and bn is not None and bn.training and bias is None and not unit.transposed and unit.stride == 1 and C == 32):
            # the statistics' block sums come out of the convolution's own epilogue (32-channel stride-1 units: 6 of PSMNet's 25)
            fused = ops.conv3d_k3_bnstats(x, wp_fwd, C)
            if fused is not None:
                raw, partials = fused
        if partials is None:
            raw = _conv_raw(unit, x, wp_fwd, bias)
        code = _relu_code(relu)
        y, mean, invstd, scale, shift, batch_stats = _unit_bn_forward(bn, raw, gamma, beta, skip, code, C, x.device, partials)
        ctx.unit, ctx.code, ctx.batch_stats = unit, code, batch_stats
        ctx.has = (bias is not None, gamma is not None, beta is not None, skip is not None)
        # (the data-gradient pack belongs to the weight version of THIS forward: autograd forbids changing the weight before the
        # backward pass, and the group only re-packs when a version moved)
        ctx.wp_bwd, ctx.w_version = wp_bwd, weight._version
        ctx.save_for_backward(x, w, raw, y if code == 1 and skip is not None else None, scale, shift, mean, invstd)
        return _carried_outputs(ctx, y, x_in, skip, carry_x, carry_skip)

    @staticmethod
    def backward(ctx, *grads):
        dy, dx_acc, dres_acc = _carried_grads(ctx, grads)
        x, w, raw, y, scale, shift, mean, invstd = ctx.saved_tensors
        unit, code = ctx.unit, ctx.code
        has_bias, has_gamma, has_beta, has_skip = ctx.has
        dy = torch.zeros_like(raw) if dy is None else dy.contiguous()
        need_dres = has_skip and ctx.needs_input_grad[5]
        if not need_dres:
            dres_acc = None
        dc, dgamma, dbeta, dres = ops.bn_act_bwd(dy, raw, y, scale, shift, mean, invstd, _mask_mode(code, has_skip), ctx.batch_stats,
                                                 want_dres=need_dres and code == 1, dres_acc=dres_acc)
        if need_dres and code != 1 and dres_acc is None:
            dres = dy                       # the skip branch joins after the activation (or there is none)
        dw = dx = dbias = None
        if ctx.needs_input_grad[1]:
            if unit.transposed:
                dw = ops.deconv3d_k3s2_wgrad(x, dc)
            elif unit.stride == 2:
                dw = ops.conv3d_k3s2_wgrad(x, dc)
            else:
                dw = ops.conv3d_k3_wgrad(x, dc)
        if ctx.needs_input_grad[0]:
            wp = ctx.wp_bwd if w._version == ctx.w_version else None
            dx = (ops.deconv3d_k3s2_dgrad(dc, w, residual=dx_acc, wpack=wp) if unit.transposed
                  else ops.conv3d_k3_dgrad(dc, w, unit.stride, tuple(x.shape[2:]), residual=dx_acc, wpack=wp))
        if has_bias and ctx.needs_input_grad[2]:
            # sum of dc over (batch, voxels) without another pass: dc = scale * (dpre - mean terms), so it is scale * dbeta
            # with running statistics (or no BatchNorm: scale = 1) and exactly zero behind batch statistics
            # (sum xhat = 0: the normalisation removes any constant the bias adds)
            dbias = torch.zeros_like(dbeta) if ctx.batch_stats else scale * dbeta
        return (dx, dw, dbias, dgamma if has_gamma and ctx.needs_input_grad[3] else None,
                dbeta if has_beta and ctx.needs_input_grad[4] else None, dres if need_dres else None, None, None, None, None)


def conv_unit(unit, x, residual=None, relu=False):
    """Differentiable forward of a FusedConv3d unit (``relu``: False / True = after the skip add / 'pre' = before it)."""
    conv = unit[0]
    bn = unit[1] if unit.has_bn else None
    gamma = bn.weight if bn is not None and bn.affine else None
    beta = bn.bias if bn is not None and bn.affine else None
    reg, x, residual, cx, cs = _carry_plan(x, residual)
    out = ConvUnitFn.apply(x, conv.weight, conv.bias, gamma, beta, residual, unit, relu, cx, cs)
    if not (cx or cs):
        return out
    i = 1
    if cx:
        reg[id(x)] = (x, out[i])
        i += 1
    if cs:
        reg[id(residual)] = (residual, out[i])
    return out[0]


class CatConvUnitFn(torch.autograd.Function):
    """The aggregator's FIRST unit on a concatenation / difference volume of unit disparity step (aggregators/PSMNet.py:31-35 on
    cat_fms.py:7-48, aggregators/StereoNet.py on dif_fms.py:7-46) without the volume in the forward pass: the convolution runs
    in its 2-D form on the two feature maps (csrc/catconv.hip, as the eval path does since round 2 -- a third of the layer's
    multiplications, no 400 MB volume written and read), BatchNorm / ReLU as in ConvUnitFn.  The backward pass does not build the
    volume either: the weight gradient folds z of dc into 2 x 9 maps per channel and is two 2-D weight gradients against them
    (ops.cat_first_wgrad); the data gradient -- only when the feature maps want one -- is the 3-D convolution it is, folded onto
    the maps with the builders' adjoint.  A PSMNet training step at 4 x 256x512: forward of the first unit 1.24 -> 0.2 ms, its
    weight gradient 1.28 -> 0.2 ms."""

    @staticmethod
    def forward(ctx, left, right, weight, bias, gamma, beta, unit, relu, disp_idx, kind):
        left, right = left.contiguous(), right.contiguous()
        w = weight.detach().contiguous()
        C = unit.out_planes
        sc = sh = None
        if bias is not None:
            sc, sh = _const(1.0, bias.numel(), bias.device), bias.detach().contiguous()
        raw = ops.catconv_first(left, right, len(disp_idx), ops.catconv_pack(w, kind), sc, sh, False)
        bn = unit[1] if unit.has_bn else None
        code = _relu_code(relu)
        y, mean, invstd, scale, shift, batch_stats = _unit_bn_forward(bn, raw, gamma, beta, None, code, C, left.device)
        ctx.unit, ctx.code, ctx.batch_stats, ctx.disp_idx, ctx.kind = unit, code, batch_stats, tuple(disp_idx), kind
        ctx.has = (bias is not None, gamma is not None, beta is not None)
        ctx.save_for_backward(left, right, w, raw, scale, shift, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        left, right, w, raw, scale, shift, mean, invstd = ctx.saved_tensors
        has_bias, has_gamma, has_beta = ctx.has
        idx = list(ctx.disp_idx)
        dc, dgamma, dbeta, _ = ops.bn_act_bwd(dy.contiguous(), raw, None, scale, shift, mean, invstd, _mask_mode(ctx.code, False),
                                              ctx.batch_stats)
        dw = dL = dR = dbias = None
        if ctx.needs_input_grad[2]:
            if left.shape[-1] % 4 == 0:
                dw = ops.cat_first_wgrad(left, right, dc, ctx.kind)      # z folded into 2-D maps: no volume here either
            else:
                vol = ops.cat_fms(left, right, idx) if ctx.kind == "cat" else ops.dif_fms(left, right, idx)
                dw = ops.conv3d_k3_wgrad(vol, dc)
                del vol
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dvol = ops.conv3d_k3_dgrad(dc, w, 1)
            dL, dR = ops.cat_fms_bwd(dvol, idx) if ctx.kind == "cat" else ops.dif_fms_bwd(dvol, idx)
        if has_bias and ctx.needs_input_grad[3]:
            dbias = torch.zeros_like(dbeta) if ctx.batch_stats else scale * dbeta
        return (dL if ctx.needs_input_grad[0] else None, dR if ctx.needs_input_grad[1] else None, dw, dbias,
                dgamma if has_gamma and ctx.needs_input_grad[4] else None, dbeta if has_beta and ctx.needs_input_grad[5] else None,
                None, None, None, None)


def cat_conv_unit(unit, lazy, relu=False):
    """Differentiable forward of a FusedConv3d unit on a LazyCatVolume (the caller has checked ops.catconv_applicable)."""
    conv = unit[0]
    bn = unit[1] if unit.has_bn else None
    gamma = bn.weight if bn is not None and bn.affine else None
    beta = bn.bias if bn is not None and bn.affine else None
    return CatConvUnitFn.apply(lazy.reference_fm, lazy.target_fm, conv.weight, conv.bias, gamma, beta, unit, relu,
                               tuple(lazy.disp_idx), lazy.kind)


class HeadConvFn(torch.autograd.Function):
    """nn.Conv3d(C, 1, 3, 1, 1) (+ bias) (+ the cumulative cost add of PSMNet.py:71-72)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual):
        x = x.contiguous()
        w = weight.detach().contiguous()
        ctx.save_for_backward(x, w)
        ctx.has = (bias is not None, residual is not None)
        y = ops.conv3d_k3_c1(x, w, 0.0, residual)
        # the kernel takes its bias as a host scalar; reading the parameter back every step would stall the stream, so the
        # 1-channel result gets it from the device tensor
        return y + bias.detach().view(1, 1, 1, 1, 1) if bias is not None else y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = dbias = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv3d_k3_dgrad(dy, w, 1)              # a convolution 1 -> C on the mirrored taps
        if ctx.needs_input_grad[1]:
            dw = ops.conv3d_k3_wgrad(x, dy)
        if ctx.has[0] and ctx.needs_input_grad[2]:
            dbias = dy.sum().reshape(1)
        return dx, dw, dbias, (dy if ctx.has[1] and ctx.needs_input_grad[3] else None)


class HeadDeconvFn(torch.autograd.Function):
    """nn.ConvTranspose3d(C, Co <= 32, 3, stride 2, padding 1, output_padding 1, bias) without BatchNorm / activation: GC-Net's
    output layer (aggregators/GCNet.py:63-67)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        w = weight.detach().contiguous()
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return ops.deconv3d_k3s2(x, ops.pack_deconv3d_weights(w), w.shape[1], None,
                                 bias.detach().float().contiguous() if bias is not None else None, None, False)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        return (ops.deconv3d_k3s2_dgrad(dy, w) if ctx.needs_input_grad[0] else None,
                ops.deconv3d_k3s2_wgrad(x, dy) if ctx.needs_input_grad[1] else None,
                dy.sum(dim=(0, 2, 3, 4)) if ctx.has_bias and ctx.needs_input_grad[2] else None)


class CatFmsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, left, right, disp_idx):
        ctx.disp_idx = tuple(disp_idx)
        return ops.cat_fms(left, right, list(disp_idx))

    @staticmethod
    def backward(ctx, dvol):
        dL, dR = ops.cat_fms_bwd(dvol.contiguous(), list(ctx.disp_idx))
        return dL, dR, None


class DifFmsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, left, right, disp_idx):
        ctx.disp_idx = tuple(disp_idx)
        return ops.dif_fms(left, right, list(disp_idx))

    @staticmethod
    def backward(ctx, dvol):
        dL, dR = ops.dif_fms_bwd(dvol.contiguous(), list(ctx.disp_idx))
        return dL, dR, None


FAST_FMS_BWD_MAX_W = 1024   # dmb_fast_fms_bwd_f32: 2 * 8 channels * W * 4 bytes of LDS <= 64 KiB


def fast_samples_for_autograd(disp_sample, like):
    """The samples as FastFmsFn takes them (FP32, contiguous, on the features' device) without cutting their autograd history."""
    if disp_sample.requires_grad and torch.is_grad_enabled() and disp_sample.dim() != 4:
        raise NotImplementedError("fast_cat_fms / fast_dif_fms: a gradient for the samples needs per-pixel samples [B, D, H, W]")
    return disp_sample.float().to(like.device).contiguous()


class FastFmsFn(torch.autograd.Function):
    """fast_cat_fms / fast_dif_fms under autograd (csrc/warp_volume.hip): gradients for the two feature maps (the sampler's
    adjoint) and, when they require one, for per-pixel disparity samples (the sampler's column derivative: the route AnyNet's
    and DeepPruner's predicted disparities take, AnyNet.py:60-73, DeepPruner.py:192).  ``normalize`` = fast_dif_fms's p-norm
    over the channels (dif_fms.py:82-84)."""

    @staticmethod
    def forward(ctx, left, right, disp_sample, dif, normalize=False, p=1.0):
        if left.shape[-1] > FAST_FMS_BWD_MAX_W:   # refuse here, not inside backward(): the adjoint keeps 2 x 8 rows of W floats in LDS
            raise NotImplementedError("fast_cat_fms / fast_dif_fms under autograd: feature maps wider than %d columns have no "
                                      "backward on the HIP path (the forward alone has no such limit: run it under torch.no_grad())"
                                      % FAST_FMS_BWD_MAX_W)
        if normalize and not dif:
            raise ValueError("normalize belongs to fast_dif_fms")
        ctx.dif, ctx.p = bool(dif), float(p)
        out = ops.fast_dif_fms(left, right, disp_sample, normalize, p) if dif else ops.fast_cat_fms(left, right, disp_sample)
        ctx.save_for_backward(left, right, disp_sample, out if normalize else None)
        return out

    @staticmethod
    def backward(ctx, dvol):
        left, right, disp_sample, norm_out = ctx.saved_tensors
        dL, dR, dS = ops.fast_fms_bwd(left, right, disp_sample, dvol.contiguous(), ctx.dif, norm_out, ctx.p,
                                      wrt_samples=ctx.needs_input_grad[2])
        return dL, dR, dS, None, None, None


class UpsampleRegressFn(torch.autograd.Function):
    """(cost [B, Do, Ho, Wo], disp [B, 1, Ho, Wo]) of a low-resolution cost [B, Di, Hi, Wi].  Differentiable through the
    disparity (the path every PSMNet loss takes, without the full-size gradient volume) and through the volume itself."""

    @staticmethod
    def forward(ctx, x, size, values, alpha):
        x = x.contiguous()
        cost, disp = ops.trilinear_ac_soft_argmin(x, size, values, alpha)
        ctx.save_for_backward(x, disp)
        ctx.size, ctx.values, ctx.alpha = tuple(size), tuple(values), float(alpha)
        ctx.set_materialize_grads(False)
        return cost, disp

    @staticmethod
    def backward(ctx, dcost, ddisp):
        if dcost is None and ddisp is None:
            return None, None, None, None
        x, disp = ctx.saved_tensors
        if ddisp is None:   # only a loss on the volume
            return ops.trilinear_ac_bwd(dcost.contiguous(), tuple(x.shape[1:])), None, None, None
        return ops.trilinear_ac_soft_argmin_bwd(x, disp, ddisp.contiguous(), ctx.size, list(ctx.values), ctx.alpha,
                                                grad_cost=dcost.contiguous() if dcost is not None else None), None, None, None


class DeconvK8S4Fn(torch.autograd.Function):
    """AcfNet's learned up-sampling nn.ConvTranspose3d(1, 1, 8, 4, 2) on a squeezed cost [B, D, H, W] (AcfNet.py:55-57,81-83)."""

    @staticmethod
    def forward(ctx, x, weight):
        x = x.contiguous()
        w = weight.detach().contiguous().view(8, 8, 8)
        ctx.save_for_backward(x, w)
        ctx.wshape = tuple(weight.shape)
        return ops.deconv3d_k8s4_c1(x, w)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx, dw = ops.deconv3d_k8s4_c1_bwd(x, w, dy.contiguous(), ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dx, (dw.view(ctx.wshape) if dw is not None else None)


class ConfHeadFn(torch.autograd.Function):
    """Confidence logits of one AcfNet head (cmn/cmn.py:21-36): Conv2d(D -> D/3, 3x3) + BatchNorm2d + ReLU + Conv2d(-> 1, 1x1),
    no biases.  Forward and backward are 2-D launches of the same kernel families as the 3-D units."""

    @staticmethod
    def forward(ctx, cost, w1, gamma, beta, w2, head):
        cost = cost.contiguous()
        w1d, w2d = w1.detach().contiguous(), w2.detach().contiguous()
        Cm = w1d.shape[0]
        raw = ops.conv2d(cost, ops.pack_conv2d_weights(w1d), Cm, 3)
        bn = head.conf_net[0][1] if head.batch_norm else None
        h, mean, invstd, scale, shift, batch_stats = _unit_bn_forward(bn, raw, gamma, beta, None, 1, Cm, cost.device)
        logit = ops.conv2d(h, ops.pack_conv2d_weights(w2d), 1, 1)
        ctx.batch_stats = batch_stats
        ctx.has = (gamma is not None, beta is not None)
        ctx.save_for_backward(cost, w1d, w2d, raw, h, scale, shift, mean, invstd)
        return logit

    @staticmethod
    def backward(ctx, dlogit):
        cost, w1, w2, raw, h, scale, shift, mean, invstd = ctx.saved_tensors
        dlogit = dlogit.contiguous()
        dh = ops.conv2d_dgrad(dlogit, w2)                                  # 1 -> Cm channels
        dw2 = ops.channel_dot(h, dlogit).view_as(w2) if ctx.needs_input_grad[4] else None
        dc, dgamma, dbeta, _ = ops.bn_act_bwd(dh, raw, None, scale, shift, mean, invstd, "pre", ctx.batch_stats)   # mask from raw (_mask_mode)
        dw1 = ops.conv2d_k3_wgrad(cost, dc) if ctx.needs_input_grad[1] else None
        dcost = ops.conv2d_dgrad(dc, w1) if ctx.needs_input_grad[0] else None
        return (dcost, dw1, dgamma if ctx.has[0] and ctx.needs_input_grad[2] else None,
                dbeta if ctx.has[1] and ctx.needs_input_grad[3] else None, dw2, None)


def _bn_forward(bn, training, raw, gamma, beta, C, device):
    """Shared by the 2-D / 3-D units: (mean, invstd, scale, shift, batch_stats) for this call, running buffers updated.
    Batch statistics are used when the BatchNorm module ITSELF is in training mode (as nn.BatchNorm decides): a model in
    train() whose BatchNorm layers were put in eval() -- fine-tuning with frozen statistics -- normalises with the running
    buffers and leaves them alone."""
    del training   # the enclosing unit's flag is not what decides
    batch_stats = bn is not None and bn.training
    if batch_stats:
        if bn.momentum is None:   # nn.BatchNorm: cumulative moving average, factor 1 / (batches seen including this one)
            momentum = 1.0 / float(int(bn.num_batches_tracked) + 1) if bn.track_running_stats else 0.0
        else:
            momentum = bn.momentum
        mean, invstd, scale, shift = ops.bn_train_stats(raw, gamma.detach() if gamma is not None else None,
                                                        beta.detach() if beta is not None else None,
                                                        bn.running_mean if bn.track_running_stats else None,
                                                        bn.running_var if bn.track_running_stats else None, momentum, bn.eps)
        if bn.track_running_stats and bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1
    elif bn is not None:
        mean = bn.running_mean.detach().float()
        invstd = torch.rsqrt(bn.running_var.detach().float() + bn.eps)
        scale = (gamma.detach() if gamma is not None else torch.ones_like(mean)) * invstd
        shift = (beta.detach() if beta is not None else torch.zeros_like(mean)) - mean * scale
    else:
        mean = shift = _const(0.0, C, device)
        invstd = scale = _const(1.0, C, device)
    return mean, invstd, scale, shift, batch_stats


def _as3d_weight(w):
    """[Co, Ci, 3, 3] -> [Co, Ci, 3, 3, 3] with the 2-D taps on the middle plane: a 2-D layer seen by the 3-D stride-2 kernels."""
    w3 = torch.zeros(w.shape[:2] + (3, 3, 3), dtype=w.dtype, device=w.device)
    w3[:, :, 1] = w
    return w3


# Weight packs of the 2-D units across the two views of a training step.  The backbone runs the left and the right view through the
# same modules one after the other (backbones/PSMNet.py:119-131), so every unit packed its forward weights twice per step and
# built its data-gradient weights (a transpose, a flip, a copy, a pack) twice.  A pack now lives for a PHASE: the forward pack from
# the first forward call until the unit's next backward call, the data-gradient packs from the first backward call until the next
# forward call.  Parameters only change between a backward phase and the next forward phase (whatever the optimizer does to
# ``_version``: see _PackGroup), so nothing stale survives; the version and the parameter epoch are part of the key for the
# updates that do not follow a backward pass (load_state_dict).
def _phase_pack(unit, w, forward):
    key = (w.data_ptr(), tuple(w.shape), w._version, ops.param_epoch())
    mine, other = ("_dmb_pack_fwd", "_dmb_pack_bwd") if forward else ("_dmb_pack_bwd", "_dmb_pack_fwd")
    d = unit.__dict__
    d[other] = None
    c = d.get(mine)
    if c is None or c[0] != key or not _pack_group_enabled or torch.cuda.is_current_stream_capturing():
        c = (key, ops.pack_conv2d_weights(w) if forward else ops.conv2d_dgrad_packs(w))
        d[mine] = c
    return c[1]


class Conv2dUnitFn(torch.autograd.Function):
    """y = act(BN(conv2d(x)) (+ skip)) of one FusedConv2d unit of the 2-D networks (layers/basic_layers.py:31-46,105-123):
    kernel 1 | 3, stride 1 | 2, dilation 1 | 2.  Stride-1 gradients run on the 2-D kernels; the few stride-2 layers borrow the
    3-D stride-2 kernels with a depth of one."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, skip, unit, relu, carry_x=False, carry_skip=False):
        x_in = x
        x = x.contiguous()
        w = weight.detach().contiguous()
        C, k, s, d = unit.out_planes, unit.kernel_size, unit.stride, unit.dilation
        if k == 5:
            if s != 2 or d != 1 or x.shape[2] % 2 or x.shape[3] % 2:
                raise NotImplementedError("training path of FusedConv2d: 5x5 layers are stride 2 on even input sizes")
        elif k not in (1, 3) or d not in (1, 2, 4, 8) or (s == 2 and d != 1):
            raise NotImplementedError("training path of FusedConv2d: kernel 1|3, dilation 1|2|4|8, stride 2 only without dilation")
        sc = sh = None
        if bias is not None:
            sc, sh = _const(1.0, bias.numel(), bias.device), bias.detach().contiguous()
        raw = ops.conv2d(x, _phase_pack(unit, w, True), C, k, s, d, sc, sh, None, False)
        bn = unit[1] if unit.has_bn else None
        code = _relu_code(relu)
        y, mean, invstd, scale, shift, batch_stats = _unit_bn_forward(bn, raw, gamma, beta, skip, code, C, x.device)
        ctx.unit, ctx.code, ctx.batch_stats = unit, code, batch_stats
        ctx.has = (bias is not None, gamma is not None, beta is not None, skip is not None)
        ctx.save_for_backward(x, w, raw, y if code == 1 and skip is not None else None, scale, shift, mean, invstd)
        return _carried_outputs(ctx, y, x_in, skip, carry_x, carry_skip)

    @staticmethod
    def backward(ctx, *grads):
        dy, dx_acc, dres_acc = _carried_grads(ctx, grads)
        x, w, raw, y, scale, shift, mean, invstd = ctx.saved_tensors
        unit, code = ctx.unit, ctx.code
        unit.__dict__["_dmb_pack_fwd"] = None     # any backward call ends the forward phase of the unit's packs (_phase_pack)
        k, s, d = unit.kernel_size, unit.stride, unit.dilation
        has_bias, has_gamma, has_beta, has_skip = ctx.has
        dy = torch.zeros_like(raw) if dy is None else dy.contiguous()
        need_dres = has_skip and ctx.needs_input_grad[5]
        if not need_dres:
            dres_acc = None
        dc, dgamma, dbeta, dres = ops.bn_act_bwd(dy, raw, y, scale, shift, mean, invstd, _mask_mode(code, has_skip), ctx.batch_stats,
                                                 want_dres=need_dres and code == 1, dres_acc=dres_acc)
        if need_dres and code != 1 and dres_acc is None:
            dres = dy
        dw = dx = dbias = None
        if ctx.needs_input_grad[1]:
            if k == 5:     # 5x5 stride 2 = 3x3 stride 1 on the space-to-depth input (GC-Net's first layer)
                dw = _k3_as_k5s2_grad(ops.conv2d_wgrad(_space_to_depth(x).contiguous(), dc, 3, 1), x.shape[1])
            elif s == 1:
                dw = ops.conv2d_wgrad(x, dc, k, d)
            elif k == 3:   # stride 2: the 3-D stride-2 weight gradient at depth 1, middle plane of its taps
                dw = ops.conv3d_k3s2_wgrad(x.unsqueeze(2), dc.unsqueeze(2))[:, :, 1].contiguous()
            else:          # 1x1, stride 2: a 1x1 layer on the even positions
                dw = ops.conv2d_wgrad(x[:, :, ::2, ::2].contiguous(), dc, 1, 1)
        if ctx.needs_input_grad[0]:
            if k == 5:
                dx = _depth_to_space(ops.conv2d_dgrad(dc, _k5s2_as_k3(w))).contiguous()
            elif s == 1:
                dx = ops.conv2d_dgrad(dc, w, d, residual=dx_acc, packs=_phase_pack(unit, w, False))
                dx_acc = None
            elif k == 3:
                dx = ops.conv3d_k3_dgrad(dc.unsqueeze(2), _as3d_weight(w), 2, (1,) + tuple(x.shape[2:])).squeeze(2)
            else:
                dx = torch.zeros_like(x)
                dx[:, :, ::2, ::2] = ops.conv2d_dgrad(dc, w, 1)
            if dx_acc is not None:   # the forms without a skip operand in their data-gradient launch
                dx = dx + dx_acc
        if has_bias and ctx.needs_input_grad[2]:
            dbias = torch.zeros_like(dbeta) if ctx.batch_stats else scale * dbeta
        return (dx, dw, dbias, dgamma if has_gamma and ctx.needs_input_grad[3] else None,
                dbeta if has_beta and ctx.needs_input_grad[4] else None, dres if need_dres else None, None, None, None, None)


def conv2d_unit(unit, x, residual=None, relu=False):
    conv = unit[0]
    bn = unit[1] if unit.has_bn else None
    gamma = bn.weight if bn is not None and bn.affine else None
    beta = bn.bias if bn is not None and bn.affine else None
    reg, x, residual, cx, cs = _carry_plan(x, residual)
    out = Conv2dUnitFn.apply(x, conv.weight, conv.bias, gamma, beta, residual, unit, relu, cx, cs)
    if not (cx or cs):
        return out
    i = 1
    if cx:
        reg[id(x)] = (x, out[i])
        i += 1
    if cs:
        reg[id(residual)] = (residual, out[i])
    return out[0]


class BareConv1x1Fn(torch.autograd.Function):
    """nn.Conv2d(C, Co, 1, bias=False) (the last layer of the PSMNet backbone, backbones/PSMNet.py:61-62)."""

    @staticmethod
    def forward(ctx, x, weight):
        x = x.contiguous()
        w = weight.detach().contiguous()
        ctx.save_for_backward(x, w)
        return ops.conv2d(x, ops.pack_conv2d_weights(w), w.shape[0], 1)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        return (ops.conv2d_dgrad(dy, w) if ctx.needs_input_grad[0] else None,
                ops.conv2d_wgrad(x, dy, 1, 1) if ctx.needs_input_grad[1] else None)


def _space_to_depth(x):
    """[B, C, H, W] (even H, W) -> [B, 4C, H/2, W/2], channel order (c, row parity, column parity)."""
    B, C, H, W = x.shape
    return x.view(B, C, H // 2, 2, W // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(B, 4 * C, H // 2, W // 2)


def _depth_to_space(x):
    B, C4, H2, W2 = x.shape
    C = C4 // 4
    return x.view(B, C, 2, 2, H2, W2).permute(0, 1, 4, 2, 5, 3).reshape(B, C, 2 * H2, 2 * W2)


def _k5s2_as_k3(w):
    """A 5x5 stride-2 convolution (padding 2) IS a 3x3 stride-1 convolution (padding 1) on the space-to-depth input:
    x[2(o + A - 1) + r] meets w[2A + r] (A in 0..2, r in 0..1; the sixth tap is zero).  [Co, Ci, 5, 5] -> [Co, 4Ci, 3, 3]."""
    Co, Ci = w.shape[:2]
    w6 = torch.nn.functional.pad(w, (0, 1, 0, 1))
    return w6.view(Co, Ci, 3, 2, 3, 2).permute(0, 1, 3, 5, 2, 4).reshape(Co, 4 * Ci, 3, 3).contiguous()


def _k3_as_k5s2_grad(dw3, Ci):
    Co = dw3.shape[0]
    return dw3.view(Co, Ci, 2, 2, 3, 3).permute(0, 1, 4, 2, 5, 3).reshape(Co, Ci, 6, 6)[:, :, :5, :5].contiguous()


class BareConv2dFn(torch.autograd.Function):
    """A bare nn.Conv2d with bias (backbones/StereoNet.py:26-27,76: 5x5 stride 2 or 3x3 stride 1; edge_aware.py:42 with its skip
    add and ReLU): y = act(conv(x) + bias (+ skip)).  The 5x5 stride-2 layers take their gradients through the equivalent 3x3
    stride-1 layer on the space-to-depth input."""

    @staticmethod
    def forward(ctx, x, weight, bias, skip, relu):
        x = x.contiguous()
        w = weight.detach().contiguous()
        Co, k = w.shape[0], w.shape[2]
        stride = 2 if k == 5 else 1
        if k == 5 and (x.shape[2] % 2 or x.shape[3] % 2):
            raise NotImplementedError("training path of the 5x5 stride-2 layers: even input sizes")
        b = bias.detach().contiguous() if bias is not None else None
        raw = ops.conv2d(x, ops.pack_conv2d_weights(w), Co, k, stride, 1, None, b, None, False)
        one, zero = _const(1.0, Co, x.device), _const(0.0, Co, x.device)
        if relu:
            y = ops.bn_act(raw, one, zero, skip, True)
        elif skip is not None:
            y = ops.bn_act(raw, one, zero, skip, False)
        else:
            y = raw
        ctx.relu, ctx.has = bool(relu), (bias is not None, skip is not None)
        ctx.save_for_backward(x, w, raw, y if relu and skip is not None else None, one, zero)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, raw, y, one, zero = ctx.saved_tensors
        has_bias, has_skip = ctx.has
        dy = dy.contiguous()
        need_dres = has_skip and ctx.needs_input_grad[3]
        dc, _, dsum, dres = ops.bn_act_bwd(dy, raw, y, one, zero, zero, one, _mask_mode(1 if ctx.relu else 0, has_skip), False,
                                           want_dres=need_dres and ctx.relu)
        if need_dres and not ctx.relu:
            dres = dy
        k, Ci = w.shape[2], w.shape[1]
        dx = dw = None
        if k == 5:
            w3 = _k5s2_as_k3(w)
            if ctx.needs_input_grad[1]:
                dw = _k3_as_k5s2_grad(ops.conv2d_wgrad(_space_to_depth(x).contiguous(), dc, 3, 1), Ci)
            if ctx.needs_input_grad[0]:
                dx = _depth_to_space(ops.conv2d_dgrad(dc, w3)).contiguous()
        else:
            if ctx.needs_input_grad[1]:
                dw = ops.conv2d_wgrad(x, dc, k, 1)
            if ctx.needs_input_grad[0]:
                dx = ops.conv2d_dgrad(dc, w)
        return dx, dw, (dsum if has_bias and ctx.needs_input_grad[2] else None), (dres if need_dres else None), None


class BilinearScaleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, out_hw, mult):
        x = x.contiguous()
        ctx.hw, ctx.mult = tuple(x.shape[2:]), float(mult)
        return ops.bilinear_scale(x, tuple(out_hw), mult)

    @staticmethod
    def backward(ctx, dy):
        return ops.bilinear_scale_bwd(dy.contiguous(), ctx.hw, ctx.mult), None, None


class AvgPool2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k):
        x = x.contiguous()
        ctx.k, ctx.hw = int(k), tuple(x.shape[2:])
        return ops.avgpool2d(x, k)

    @staticmethod
    def backward(ctx, dy):
        return ops.avgpool2d_bwd(dy.contiguous(), ctx.hw, ctx.k), None


class BilinearAcFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, out_hw):
        x = x.contiguous()
        ctx.hw = tuple(x.shape[2:])
        return ops.bilinear_ac(x, tuple(out_hw))

    @staticmethod
    def backward(ctx, dy):
        return ops.bilinear_ac_bwd(dy.contiguous(), ctx.hw), None


class SoftArgminFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cost, values, alpha):
        cost = cost.contiguous()
        disp = ops.soft_argmin(cost, list(values), alpha, True)
        ctx.save_for_backward(cost, disp)
        ctx.values, ctx.alpha = tuple(values), float(alpha)
        return disp

    @staticmethod
    def backward(ctx, ddisp):
        cost, disp = ctx.saved_tensors
        return ops.soft_argmin_bwd(cost, disp, ddisp.contiguous(), list(ctx.values), ctx.alpha), None, None


def wants_grad(module, *tensors):
    """True when a forward call must take the unit-by-unit path of this file instead of the fused inference kernels:
    (a) the module is in training mode (batch statistics and running-buffer updates are training-mode semantics with or
    without autograd, e.g. a validation pass under torch.no_grad() inside train()); or (b) gradients are enabled and
    something on this call can receive one -- an input that already carries a gradient, or one of the module's OWN
    parameters (``model.train(); model.backbone.eval()`` freezes the backbone's BatchNorm statistics, not its weights: with
    the reference's plain nn modules those weights still get gradients, and so they do here; each BatchNorm follows its own
    ``training`` flag).  Under torch.no_grad() -- how the reference's tools run evaluation, and what
    GeneralizedStereoModel's eval branch enters itself -- an eval-mode module always stays on the fused kernels."""
    if module.training:
        return True
    if not torch.is_grad_enabled():
        return False
    if any(t is not None and getattr(t, "requires_grad", False) for t in tensors):
        return True
    if any(p.requires_grad for p in module.parameters()):
        _warn_eval_on_training_path(module)
        return True
    return False


_warned_eval_path = [False]


def _warn_eval_on_training_path(module):
    """An eval-mode module called with gradients enabled and trainable parameters takes the unit-by-unit autograd path (full-size
    activations saved, no fused volume / regression kernels): several times the memory and latency of the inference path.
    That is what the reference's nn modules would do too, but it is easy to hit by accident -- say so once."""
    if not _warned_eval_path[0]:
        _warned_eval_path[0] = True
        import warnings
        warnings.warn("%s is in eval mode but gradients are enabled and its parameters require grad: taking the autograd "
                      "(training) path.  Wrap inference in torch.no_grad() -- as GeneralizedStereoModel's eval branch does -- to "
                      "stay on the fused inference kernels." % type(module).__name__, stacklevel=3)
