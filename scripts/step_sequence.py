"""Development aid: the kernels of ONE step of bench.py in launch order with their durations, from a rocprofv3 --kernel-trace CSV.
    rocprofv3 --kernel-trace -d DIR -o trace --output-format csv -- python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 2
    python scripts/step_sequence.py DIR/**/trace_kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("void ", "").replace("dmb::", "")
    n = n.split("(")[0]
    return n[:110]


# one step = from one catconv_finalize_kernel (first layer) to the next
starts = [i for i, r in enumerate(rows) if "catconv_finalize" in r["Kernel_Name"]]
if len(starts) < 2:
    starts = [0, len(rows)]
a, b = starts[-2], starts[-1]
# back up to the first conv2d of the step
while a > 0 and ("conv2d_kernel" in rows[a - 1]["Kernel_Name"] or "copy_window" in rows[a - 1]["Kernel_Name"]):
    a -= 1
    b -= 0
t_first = int(rows[a]["Start_Timestamp"])
prev_end = None
total = 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    total += (e - s) / 1e3
    print("%9.1f us  +%6.1f gap  %8.1f us  %s" % ((s - t_first) / 1e3, gap, (e - s) / 1e3, short(r["Kernel_Name"])))
    prev_end = e
print("step: %d kernels, sum of durations %.1f us, span %.1f us" % (b - a, total, (prev_end - t_first) / 1e3))
