"""Development aid: the kernels of ONE training step (scripts/train_bench.py) in launch order with durations and gaps, from a
rocprofv3 --kernel-trace CSV.
    python scripts/train_step_sequence.py DIR/**/trace_kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("void ", "").replace("dmb::", "").replace("at::native::", "")
    n = n.split("(")[0]
    return n[:120]


# one step = from one marker launch to the next: the first layer's weight pack (once per forward pass) where the training path runs
# its first unit without the volume, else the cat volume
marker = "catconv_pack_kernel" if any("catconv_pack_kernel" in r["Kernel_Name"] for r in rows) else "volume_kernel"
starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"] and "bwd" not in r["Kernel_Name"]]
a, b = starts[-3], starts[-2]
t_first = int(rows[a]["Start_Timestamp"])
prev_end = None
total = 0.0
gaps = 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    gaps += max(gap, 0.0)
    total += (e - s) / 1e3
    print("%9.1f us  +%6.1f gap  %8.1f us  grid %-8s %s" % ((s - t_first) / 1e3, gap, (e - s) / 1e3, r.get("Grid_Size_X", ""), short(r["Kernel_Name"])))
    prev_end = e
print("step: %d kernels, sum of durations %.1f us, sum of gaps %.1f us, span %.1f us" % (b - a, total, gaps, (prev_end - t_first) / 1e3))
