"""Data-parallel gradient exchange of the training side (last part): the role of the reference's
``all_reduce_grads`` / ``DistOptimizerHook`` (dmb/utils/dist_utils.py:16-64) on RCCL over xGMI.

The reference flattens the gradients into type buckets after every backward pass, all-reduces the copies and copies
them back.  Here one flat FP32 buffer per model is exchanged with ONE collective -- the PSMNet cost path's 5.2 M parameters
are 20.9 MB, far below the size at which splitting a ring all-reduce over the seven xGMI links of an MI355X would pay for
its extra launches -- and afterwards every ``param.grad`` IS a view into it: nothing is copied back.  How the gradients get
into the buffer is a mode of FlatGradients:

  "gather" (default, round 6)  ``zero_()`` drops the gradients (``grad = None``, no launch); autograd then hands each parameter
        the tensor its backward kernel wrote (no accumulation launch: rounds 1-5 paid one ``grad += new`` per parameter,
        75 launches of 4 us in a PSMNet step); ``all_reduce()`` packs them with one multi-tensor copy.  With one rank nothing is
        packed at all.
  "accumulate"                 every ``param.grad`` is a view from the start, ``zero_()`` clears the buffer with one fill and
        autograd accumulates straight into the views: what gradient accumulation over several backward passes wants.

Averaging is part of the collective where the backend offers it (RCCL: ncclAvg), one in-place scale otherwise (gloo, used by
the CPU tests).

``torch.distributed`` is plumbing here (process group, RCCL); nothing in this file computes on the data path.
"""
import torch
import torch.distributed as dist

__all__ = ["FlatGradients", "all_reduce_grads"]


class FlatGradients(object):
    """One contiguous gradient buffer for all learnable parameters of ``model``; ``mode`` "gather" | "accumulate" (module docstring)."""

    def __init__(self, model, mode="gather"):
        if mode not in ("gather", "accumulate"):
            raise ValueError("FlatGradients: mode must be 'gather' or 'accumulate'")
        self.mode = mode
        self.params = [p for p in model.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGradients: the model has no learnable parameter")
        first = self.params[0]
        if any(p.dtype != first.dtype or p.device != first.device for p in self.params):
            raise ValueError("FlatGradients: parameters must share one dtype and one device")
        # 16-byte aligned slots so that every view can be read with vector loads by the optimizer kernels
        self.offsets, total = [], 0
        for p in self.params:
            self.offsets.append(total)
            total += (p.numel() + 3) // 4 * 4
        self.flat = torch.zeros(total, dtype=first.dtype, device=first.device)
        self.views = [self.flat[o:o + p.numel()].view_as(p) for p, o in zip(self.params, self.offsets)]
        if mode == "accumulate":
            for p, v in zip(self.params, self.views):
                p.grad = v
        model._dmb_flat_grads = self

    def zero_(self):
        """Replaces optimizer.zero_grad().  "gather": the gradients are dropped (None) -- the next backward pass hands every
        parameter a fresh tensor.  "accumulate": keeps the views, clears the buffer with one fill.
        NOTE ("accumulate"): every parameter keeps a (zero) ``.grad`` -- a parameter that takes no part in a step still gets an
        optimizer update with a zero gradient (weight decay, Adam moments), where the reference leaves its ``.grad`` None and the
        optimizer skips it.  The cost path has no such parameter; freeze one with ``requires_grad_(False)`` BEFORE building
        FlatGradients (frozen parameters are left out of the buffer) if that difference matters."""
        if self.mode == "gather":
            for p in self.params:
                p.grad = None
            return self
        self.flat.zero_()
        for p, v in zip(self.params, self.views):   # re-attach views an optimizer.zero_grad(set_to_none=True) dropped
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v
        return self

    def attached(self):
        return all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(self.params, self.views))

    def gather_(self):
        """Pack the gradients autograd left on the parameters into the buffer (one multi-tensor copy) and make every
        ``param.grad`` its view; a parameter without a gradient contributes zeros (the other ranks may have one)."""
        src, dst = [], []
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None:
                v.zero_()
            elif g.data_ptr() != v.data_ptr():
                src.append(g.detach())
                dst.append(v)
            p.grad = v
        if src:
            torch._foreach_copy_(dst, src)
        return self

    def all_reduce(self, group=None, async_op=False):
        """Average over the ranks of ``group`` in place; returns the work handle when ``async_op``."""
        world = dist.get_world_size(group)
        if world == 1:
            return None
        if dist.get_backend(group) != "nccl" and async_op:   # checked BEFORE anything is launched: no half-done, un-averaged collective is left in flight
            raise ValueError("FlatGradients.all_reduce: async_op needs a backend with an averaging reduction (RCCL)")
        if not self.attached():
            self.gather_()
        if dist.get_backend(group) == "nccl":   # RCCL
            return dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=group, async_op=async_op)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.flat.div_(world)
        return None


def all_reduce_grads(model, coalesce=True, bucket_size_mb=-1):
    """Same call as the reference's ``all_reduce_grads`` (dist_utils.py:36-48): average ``param.grad`` over all ranks.
    With a FlatGradients attached this is one collective on the buffer itself ("accumulate" mode: while its views are still in
    place); otherwise
    the gradients are packed into one temporary buffer (``coalesce``) or reduced one by one.  ``bucket_size_mb`` is
    accepted for source compatibility: a single bucket is the right size on xGMI for this model (module docstring)."""
    flat = getattr(model, "_dmb_flat_grads", None)
    if flat is not None and (flat.mode == "gather" or flat.attached()):
        flat.all_reduce()
        return
    grads = [p.grad.data for p in model.parameters() if p.requires_grad and p.grad is not None]
    world = dist.get_world_size()
    if world == 1 or not grads:
        return
    if not coalesce:
        for g in grads:
            dist.all_reduce(g)
            g.div_(world)
        return
    buf = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(buf)
    buf.div_(world)
    o = 0
    for g in grads:
        g.copy_(buf[o:o + g.numel()].view_as(g))
        o += g.numel()
