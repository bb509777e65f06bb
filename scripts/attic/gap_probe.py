"""Development aid: back-to-back launches of one kernel at several grid sizes under `rocprofv3 --kernel-trace`, to see what the
~5 us dispatch gaps on either side of the full-resolution stride-1 launches depend on (kernel, grid size, bytes written).
    rocprofv3 --kernel-trace -d DIR -o t --output-format csv -- python scripts/attic/gap_probe.py ; python scripts/attic/gap_probe.py report DIR/t_kernel_trace.csv"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 2 and sys.argv[1] == "report":
    import csv
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    prev = None
    for r in rows:
        n = r["Kernel_Name"].replace("void ", "").replace("dmb::", "").split("(")[0][:70]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if "conv3d" in n or "deconv3d" in n:
            print("%-72s grid %9s  dur %8.1f us  gap before %6.1f us" % (n, r.get("Grid_Size", "?"), (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
        prev = e
    sys.exit(0)
import torch
from densematchingbenchmark_amd import ops
dev = torch.device("cuda:0")
g = lambda *s: torch.randn(*s, device=dev)
wp32 = ops.pack_conv3d_weights(g(32, 32, 3, 3, 3) * 0.03)
wp64 = ops.pack_conv3d_weights(g(64, 64, 3, 3, 3) * 0.03)
sc32, sh32 = torch.ones(32, device=dev), torch.zeros(32, device=dev)
sc64, sh64 = torch.ones(64, device=dev), torch.zeros(64, device=dev)
xs = {B: g(B, 32, 48, 136, 240) for B in (1, 2, 4)}
xh = {B: g(B, 64, 24, 68, 120) for B in (4, 16)}
torch.cuda.synchronize()
for B in (4, 2, 1):
    for _ in range(4):
        ops.conv3d_k3(xs[B], wp32, 32, sc32, sh32, None, 1, True)
for B in (4, 16):
    for _ in range(4):
        ops.conv3d_k3(xh[B], wp64, 64, sc64, sh64, None, 1, True)
for _ in range(3):
    ops.conv3d_k3(xs[4], wp32, 32, sc32, sh32, None, 1, True)
    ops.conv3d_k3(xh[4], wp64, 64, sc64, sh64, None, 1, True)
torch.cuda.synchronize()
