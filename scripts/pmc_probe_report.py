"""Per-kernel means of every counter collected by scripts/pmc_probe.sh."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(os.path.join(root, "g*", "**", "*counter_collection.csv"), recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))   # (dispatch) -> counter -> sum over dims
    names = {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dmb::", "")
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = k
    for d, cs in per.items():
        for c, v in cs.items():
            vals[names[d]][c].append(v)
for k, cs in vals.items():
    if not any(s in k for s in ("conv3d", "deconv3d", "trilinear", "volume", "soft_argmin", "gwc", "conv_c1", "conv2d_kernel")):
        continue
    print(k)
    for c, v in cs.items():
        v = v[1:] if len(v) > 1 else v      # first launch = cold
        print("   %-40s %14.6g   (n=%d)" % (c, sum(v) / len(v), len(v)))
