import hashlib
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def sha(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


def maxdiff(a, b):
    a = torch.as_tensor(np.asarray(a)) if not torch.is_tensor(a) else a.detach().cpu()
    b = torch.as_tensor(np.asarray(b)) if not torch.is_tensor(b) else b.detach().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return (a.double() - b.double()).abs().max().item()
