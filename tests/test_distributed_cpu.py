"""N > 1 path on CPU: two processes over gloo.  The HIP kernels cannot run here, so each rank fills its accumulator
rows from the oracle's per-image errors for ITS shard (pair i -> rank i mod world); the test checks the sharding,
the single SUM all-reduce and the mean-over-images summary against the oracle evaluated on the whole dataset."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from densematchingbenchmark_amd.evaluation import EpeAccumulator, local_batches, shard_indices
from oracle import dmb_oracle as O

NUM_PAIRS, WORLD = 7, 2


def _pair(i):
    g = torch.Generator().manual_seed(9000 + i)
    gt = torch.rand((1, 1, 12, 20), generator=g) * 220 - 10
    est = gt + torch.randn((1, 1, 12, 20), generator=g) * 2
    return est, gt


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    acc = EpeAccumulator(torch.device("cpu"), 1, 0, 192)
    mine = shard_indices(NUM_PAIRS, rank, world)
    for chunk in local_batches(mine, 2):
        for i in chunk:
            est, gt = _pair(i)
            e = O.calc_error(O.remove_padding(est, (10, 18)), O.remove_padding(gt, (10, 18)), 0, 192)
            acc.acc[0] += torch.tensor([1.0] + [e[k] for k in ("epe", "1px", "2px", "3px", "5px")], dtype=torch.float64)
    acc.all_reduce()
    s = acc.summary()[0]
    out.put((rank, mine, int(acc.acc[0, 0].item()), [s[k] for k in ("epe", "1px", "2px", "3px", "5px")]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_indices_partition():
    all_idx = sorted(i for r in range(4) for i in shard_indices(10, r, 4))
    assert all_idx == list(range(10)) and shard_indices(10, 1, 4) == [1, 5, 9]
    assert local_batches([0, 2, 4, 6, 8], 2) == [[0, 2], [4, 6], [8]]


def test_two_rank_all_reduce_matches_whole_dataset():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(WORLD)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ests, gts = zip(*[_pair(i) for i in range(NUM_PAIRS)])
    ref, n = O.dataset_metrics(list(ests), list(gts), (10, 18), 0, 192)
    shards = sorted(i for _, mine, _, _ in got for i in mine)
    assert shards == list(range(NUM_PAIRS))
    for rank, mine, count, vals in got:
        assert count == NUM_PAIRS == n          # every rank sees the whole-job image count after the all-reduce
        assert np.allclose(vals, [ref[k] for k in ("epe", "1px", "2px", "3px", "5px")], rtol=1e-9, atol=1e-9)


def _grad_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from densematchingbenchmark_amd.dist_utils import FlatGradients, all_reduce_grads
    torch.manual_seed(0)                      # same initial weights on every rank
    model = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    res = {}
    for mode in ("flat", "flat_accumulate", "coalesced", "one_by_one"):
        model.zero_grad(set_to_none=True)
        if hasattr(model, "_dmb_flat_grads"):
            del model._dmb_flat_grads
        flat = None
        if mode.startswith("flat"):
            flat = FlatGradients(model, mode="accumulate" if mode == "flat_accumulate" else "gather").zero_()
        x = torch.randn(4, 5, generator=torch.Generator().manual_seed(100 + rank))   # this rank's shard of the batch
        model(x).square().mean().backward()
        if flat is not None:
            # "accumulate": autograd accumulated into the views; "gather": fresh tensors that the exchange packs
            assert flat.attached() == (mode == "flat_accumulate")
        all_reduce_grads(model, coalesce=(mode != "one_by_one"))
        if flat is not None:
            assert flat.attached()            # after the exchange every grad IS a view of the reduced buffer
        res[mode] = [p.grad.clone() for p in model.parameters()]
    out.put((rank, {k: [g.numpy() for g in v] for k, v in res.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_all_reduce():
    """dist_utils.all_reduce_grads (reference dmb/utils/dist_utils.py:36-48): the averaged gradient of two ranks equals
    the gradient of the mean of the two shard losses, identically on both ranks and for all four exchange modes."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, WORLD, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(WORLD))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    loss = sum(model(torch.randn(4, 5, generator=torch.Generator().manual_seed(100 + r))).square().mean() for r in range(WORLD)) / WORLD
    ref = [g.numpy() for g in torch.autograd.grad(loss, list(model.parameters()))]
    for rank in range(WORLD):
        for mode, grads in got[rank].items():
            for a, b in zip(grads, ref):
                assert np.allclose(a, b, rtol=1e-6, atol=1e-7), (rank, mode)
