"""Sharded evaluation loop: the shape of tools/test.py::multi_gpu_test (:101-170) + collect_results (:172-208)
without the pickle-file exchange.

Pair ``i`` of the dataset goes to rank ``i mod world`` (tools/test.py:108); each rank runs its pairs through the
model in local batches, folds every image into the device-side EPE accumulator, and ONE all-reduce at the end
gives every rank the dataset metrics (mean over images of per-image masked means, tools/test.py:304-307)."""
import torch

from .stereo import EpeAccumulator


def shard_indices(num_pairs, rank, world):
    """tools/test.py:108 -- ``range(rank, len(dataset), world_size)``."""
    return list(range(rank, num_pairs, world))


def local_batches(indices, batch_size):
    return [indices[i:i + batch_size] for i in range(0, len(indices), batch_size)]


def evaluate_sharded(model, make_batch, num_pairs, batch_size, rank, world, device, original_size, lower_bound,
                     upper_bound, num_ids=1):
    """``make_batch(list_of_pair_indices) -> (batch_dict, gt_disp [B,1,Hp,Wp])`` on ``device``.
    Returns (per-disparity-id metric dicts, number of images evaluated by the whole job)."""
    acc = EpeAccumulator(device, num_ids, lower_bound, upper_bound)
    with torch.no_grad():
        for idx in local_batches(shard_indices(num_pairs, rank, world), batch_size):
            batch, gt = make_batch(idx)
            results, _ = model(batch)
            acc.update(results["disps"][:num_ids], gt, original_size)
    acc.all_reduce()
    return acc.summary(), int(round(acc.acc[0, 0].item()))
