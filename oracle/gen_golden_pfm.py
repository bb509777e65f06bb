"""Generate tests/golden/pfm_files.npz: three small PFM files (gray little-endian, gray big-endian with a scale, colour) as
bytes, and what the REFERENCE's loader (dmb/data/datasets/utils/load_disp.py:5-53, imported from the reference tree) returns
for each: the array, its dtype string and the scale.  Fixtures are data (file bytes in, arrays out).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_pfm.py
"""
import importlib.util
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = os.environ.get("DMB_REFERENCE", "/root/reference")

from densematchingbenchmark_amd import disp_io  # noqa: E402  (the writer only)


def main():
    spec = importlib.util.spec_from_file_location("ref_load_disp", os.path.join(REF, "dmb/data/datasets/utils/load_disp.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rs = np.random.RandomState(5)
    cases = [("gray_le", rs.rand(7, 11).astype(np.float32) * 191, 1.0, True),
             ("gray_be", rs.rand(5, 4).astype(np.float32) * 50, 2.5, False),
             ("color_le", rs.rand(3, 6, 3).astype(np.float32), 1.0, True)]
    out = {}
    d = tempfile.mkdtemp()
    for name, arr, scale, le in cases:
        p = os.path.join(d, name + ".pfm")
        disp_io.write_pfm(p, arr, scale, le)
        got, s = ref.load_pfm(p)                       # the reference's own loader
        assert np.array_equal(got, arr) and s == scale, name
        if name.startswith("gray"):
            assert np.array_equal(ref.load_scene_flow_disp(p), got)
        out[name + "_bytes"] = np.frombuffer(open(p, "rb").read(), dtype=np.uint8)
        out[name + "_data"] = np.ascontiguousarray(got).astype(np.float32)
        out[name + "_scale"] = np.float64(s)
        out[name + "_dtype"] = np.array(got.dtype.str)
    out_dir = os.environ.get("DMB_GOLDEN_OUT", os.path.join(ROOT, "tests", "golden"))   # as the other generators
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "pfm_files.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
