#!/bin/bash
# Development aid: per-kernel times of images -> disparity for ONE pair (build_model(cfg) whole, the serving regime) under rocprofv3.
#   scripts/e2e_b1_profile.sh <tag> <config relative to configs/>
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cat > /tmp/e2e_b1.py <<PY
import os, sys, torch
sys.path.insert(0, "$R")
from densematchingbenchmark_amd import ops, synthetic
from densematchingbenchmark_amd.config import Config
from densematchingbenchmark_amd.modeling import build_model
cfg = Config.fromfile("$R/configs/$2")
dev = torch.device("cuda:0")
model = build_model(cfg).eval()
synthetic.init_params_(model, seed=0, classif_gain=10.0)
model = model.to(dev)
Hp, Wp = cfg.data.eval.input_shape
l = torch.randn(1, 3, Hp, Wp, device=dev); r = torch.randn(1, 3, Hp, Wp, device=dev)
with torch.no_grad():
    for _ in range(8):
        model(dict(leftImage=l, rightImage=r))
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/e2e_$1 -o trace --output-format csv -- python /tmp/e2e_b1.py > $O/e2e_$1.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/e2e_$1/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.3f ms per call (8 calls)" % (tot / 8e6))
for r in rows[:30]:
    n = r["Name"].replace("void ", "").replace("dmb::", "").split("(")[0]
    print("%-78s calls/step %5.1f  avg %8.1f us  per step %7.1f us" % (n[:78], int(r["Calls"]) / 8, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 8e3))
PY
