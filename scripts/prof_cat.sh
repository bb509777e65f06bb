cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o t --output-format csv -- python $GRAFT_REPO_ROOT/scripts/kbench_cat.py 2>&1 | grep "2-D form"; python - <<PY
import csv,glob
f=glob.glob("/tmp/p1/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print("%-70s calls %5s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
