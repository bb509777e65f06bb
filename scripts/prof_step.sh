#!/bin/bash
# Development aid: per-kernel time of the headline step.  scripts/prof_step.sh <tag> [bench flags]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o trace --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 "$@" > $out/bench.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$out/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f ms over 7 steps" % (tot / 1e6))
for r in rows[:28]:
    n = r["Name"].replace("void ", "").replace("dmb::", "").split("(")[0]
    print("%-78s calls %4s  avg %8.1f us  %5.1f%%" % (n[:78], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
