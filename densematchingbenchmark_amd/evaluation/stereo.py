"""Evaluation side of the path: crop, per-image pixel errors, dataset mean -- on the device.

Replaces dmb/data/datasets/evaluation/stereo/{eval.py:12-31 remove_padding, pixel_error.py:6-73 calc_error} and
the pickle-file result exchange of tools/test.py:172-208,304-307 by ONE device-side accumulator of six FP64
scalars per evaluated disparity id and ONE ``all_reduce(SUM)`` over RCCL at the end.  Semantics
preserved: the dataset metric is the unweighted mean over images of the per-image masked means.
"""
import torch
import torch.distributed as dist

from .. import ops

METRIC_KEYS = ("epe", "1px", "2px", "3px", "5px")


def remove_padding(batch, size):
    """eval.py:12-37 (a view; tensors, dicts and lists)."""
    if isinstance(batch, torch.Tensor):
        pad_top = batch.shape[-2] - size[-2]
        if pad_top >= 0:
            batch = batch[:, :, pad_top:, :size[-1]]
        return batch
    if isinstance(batch, dict):
        return {k: remove_padding(v, size) for k, v in batch.items()}
    if isinstance(batch, (list, tuple)):
        return [remove_padding(v, size) for v in batch]
    raise TypeError("batch must contain tensors, dicts or lists; found {}".format(type(batch)))


class EpeAccumulator:
    """acc = [n_images, sum epe_i, sum 1px_i, sum 2px_i, sum 3px_i, sum 5px_i] per disparity id, FP64, on device."""

    def __init__(self, device, num_ids=1, lower_bound=0, upper_bound=192):
        self.acc = torch.zeros((num_ids, 6), dtype=torch.float64, device=device)
        self.lb, self.ub = float(lower_bound), float(upper_bound)

    def update(self, est_disps, gt_disp, original_size):
        """est_disps: list (one per disparity id) of PADDED [B, 1, Hp, Wp] maps; gt_disp padded the same way."""
        if isinstance(est_disps, torch.Tensor):
            est_disps = [est_disps]
        est_disps = list(est_disps)
        if 1 < len(est_disps) <= 4 and len(est_disps) == self.acc.shape[0] and all(e.shape == gt_disp.shape for e in est_disps):
            # the maps of one forward against the same ground truth: one pass instead of one per map (same sums per map)
            ops.epe_accumulate_multi(est_disps, gt_disp, self.acc, original_size, self.lb, self.ub)
            return
        for i, est in enumerate(est_disps):
            ops.epe_accumulate(est, gt_disp, self.acc[i], original_size, self.lb, self.ub)

    def all_reduce(self):
        """One 48*num_ids-byte SUM all-reduce (RCCL over xGMI on GPUs, gloo in the CPU tests)."""
        if dist.is_available() and dist.is_initialized():   # (a one-rank group reduces too: the collective still executes)
            dist.all_reduce(self.acc, op=dist.ReduceOp.SUM)
        return self

    def summary(self):
        """[{'epe': .., '1px': .., ...}] per disparity id: mean over images (tools/test.py:304-307)."""
        acc = self.acc.cpu()
        out = []
        for row in acc:
            n = max(row[0].item(), 1.0)
            out.append({k: row[1 + j].item() / n for j, k in enumerate(METRIC_KEYS)})
        return out


def calc_error(est_disp, gt_disp, lb=None, ub=None):
    """pixel_error.py:6-73 for one batch on the device: returns the per-batch MEAN over images of each metric
    (for B=1 exactly the reference's dict, as Python floats)."""
    dev = est_disp.device
    if est_disp.dim() == 2:
        est_disp, gt_disp = est_disp[None, None], gt_disp[None, None]
    elif est_disp.dim() == 3:
        est_disp, gt_disp = est_disp[:, None], gt_disp[:, None]
    acc = EpeAccumulator(dev, 1, -float("inf") if lb is None else lb, float("inf") if ub is None else ub)
    acc.update([est_disp.contiguous()], gt_disp.contiguous(), est_disp.shape[-2:])
    return acc.summary()[0]
