"""Generate tests/golden/fast_volumes_grad.npz: gradients of the REFERENCE's sample-based builders -- fast_cat_fms
(cost_processors/utils/cat_fms.py:51-82) and fast_dif_fms (dif_fms.py:49-86), imported from the reference tree -- obtained the
way the reference obtains them: torch.autograd through F.grid_sample and the expand of inverse_warp_3d.py, on CPU, for a
seeded upstream gradient.  Stored: d reference_fm, d target_fm for per-pixel samples and for the builders' own linspace
samples; and (round 4) d disp_sample for per-pixel samples that require a gradient, for the plain builders and for
fast_dif_fms(normalize=True, p in {0.5, 1, 2, 3}).  Fixtures are data.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_fast_grad.py
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402

CASES = [((2, 5, 7, 20), 6, 201), ((1, 8, 16, 33), 12, 202), ((1, 3, 4, 9), 2, 203), ((1, 16, 12, 64), 9, 204)]


def main():
    G.import_reference()
    torch.set_num_threads(8)
    from dmb.modeling.stereo.cost_processors.utils.cat_fms import fast_cat_fms
    from dmb.modeling.stereo.cost_processors.utils.dif_fms import fast_dif_fms
    warnings.filterwarnings("ignore", message="Default grid_sample")
    out = {"cases": np.array([list(sh) + [D, seed] for sh, D, seed in CASES], dtype=np.int64)}
    for i, (sh, D, seed) in enumerate(CASES):
        g = torch.Generator().manual_seed(seed + 2000)
        ds = torch.rand((sh[0], D, sh[2], sh[3]), generator=g) * sh[3] * 0.6 - 2.0      # as in gen_golden.py section 1b
        for name, fn, ch in (("cat", fast_cat_fms, 2 * sh[1]), ("dif", fast_dif_fms, sh[1])):
            for mode in ("pixel", "default"):
                a = G.rand(sh, seed).requires_grad_()
                b = G.rand(sh, seed + 1000).requires_grad_()
                vol = fn(a, b, disp_sample=ds) if mode == "pixel" else fn(a, b, 24, -3, 2)
                up = G.rand(tuple(vol.shape), seed + 3000 + (0 if name == "cat" else 1))
                assert vol.shape[1] == ch
                vol.backward(up)
                out["%s_%s_dL_%d" % (name, mode, i)] = G.npy(a.grad)
                out["%s_%s_dR_%d" % (name, mode, i)] = G.npy(b.grad)
        # round 4: the samples themselves carry a gradient (AnyNet.py:60-73 / DeepPruner.py:192 build them from predicted
        # disparities), with and without fast_dif_fms's p-norm over the channels (dif_fms.py:82-84)
        for name, fn, kw in (("cat", fast_cat_fms, {}), ("dif", fast_dif_fms, {}), ("difn1", fast_dif_fms, dict(normalize=True, p=1.0)),
                             ("difn2", fast_dif_fms, dict(normalize=True, p=2.0)), ("difn3", fast_dif_fms, dict(normalize=True, p=3.0)),
                             ("difnh", fast_dif_fms, dict(normalize=True, p=0.5))):
            a = G.rand(sh, seed).requires_grad_()
            b = G.rand(sh, seed + 1000).requires_grad_()
            s = ds.clone().requires_grad_()
            vol = fn(a, b, disp_sample=s, **kw)
            up = G.rand(tuple(vol.shape), seed + 3000 + (0 if name == "cat" else 1))
            vol.backward(up)
            out["%s_samples_dL_%d" % (name, i)] = G.npy(a.grad)
            out["%s_samples_dR_%d" % (name, i)] = G.npy(b.grad)
            out["%s_samples_dS_%d" % (name, i)] = G.npy(s.grad)
            if kw:
                out["%s_samples_out_%d" % (name, i)] = G.npy(vol)
    path = os.path.join(G.OUT, "fast_volumes_grad.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
