"""Development aid (round 5): every tile candidate of the stride-1 / stride-2 kernels computes the SAME fmaf chain per output, so on
any shape all of them must agree BIT FOR BIT (development options 19 / 10 force a candidate).  Random medium shapes -- partial
tiles in every direction, grids from under one to many workgroups per CU -- with the affine, the skip operand and the ReLU on."""
import os
os.environ.setdefault("DMB_LIB", "dev")
import random
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densematchingbenchmark_amd import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.load()
rng = random.Random(int(os.environ.get("FUZZ_SEED", "3")))
N = int(os.environ.get("FUZZ_N", "60"))
bad = 0
for it in range(N):
    B = rng.choice([1, 1, 2, 4])
    D, H = rng.randint(2, 26), rng.randint(3, 70)
    W = 4 * rng.randint(6, 80)
    stride = rng.choice([1, 1, 2])
    Co = rng.choice([32, 64]) if stride == 1 else 64
    Ci = rng.choice([32, 64]) if Co == 64 else 32
    x = torch.randn(B, Ci, D, H, W, device=dev)
    wp = ops.pack_conv3d_weights(torch.randn(Co, Ci, 3, 3, 3, device=dev) * 0.05)
    sc, sh = torch.rand(Co, device=dev) + 0.5, torch.rand(Co, device=dev) - 0.5
    do, ho, wo = [(n - 1) // stride + 1 for n in (D, H, W)]
    res = torch.randn(B, Co, do, ho, wo, device=dev) if rng.random() < 0.5 else None
    outs = []
    opts = ([(19, k) for k in range(0, 6 if Co == 64 else 5)] if stride == 1 else [(10, k) for k in (0, 1, 2, 3, 4)])
    for key, val in opts:
        lib.dmb_dev_set_option(key, val)
        outs.append(ops.conv3d_k3(x, wp, Co, sc, sh, res, stride, True))
        lib.dmb_dev_set_option(key, 0)
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    if not same:
        bad += 1
        print("MISMATCH", dict(B=B, Ci=Ci, Co=Co, D=D, H=H, W=W, stride=stride, res=res is not None),
              [float((outs[0] - o).abs().max()) for o in outs[1:]], flush=True)
print("cases %d mismatches %d" % (N, bad))
