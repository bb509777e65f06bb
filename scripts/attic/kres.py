#!/usr/bin/env python
"""Development aid: compile one csrc/*.hip for gfx950 with -Rpass-analysis=kernel-resource-usage and print a table
(kernel, VGPRs, AGPRs, SGPRs, spills, scratch, LDS, occupancy).   python scripts/attic/kres.py conv3d.hip [filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "densematchingbenchmark_amd", "csrc", sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", src,
                      "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:], capture_output=True, text=True)
if out.returncode:
    print(out.stderr[-4000:])
    sys.exit(1)
cur = None
rows = []
for line in out.stderr.splitlines():
    m = re.search(r"remark: .*?: +(.*?): +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?): +(.*?) \[-Rpass", line)
    if not m:
        m2 = re.search(r"Function Name: (\S+)", line) or re.search(r"Name: (\S+)", line)
        if m2:
            cur = {"name": m2.group(1)}
            rows.append(cur)
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k in ("Function Name", "Name"):
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
dem = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
print("%-90s %5s %5s %5s %7s %7s %8s %7s %4s" % ("kernel", "VGPR", "AGPR", "SGPR", "sSpill", "vSpill", "scratch", "LDS", "occ"))
for r, d in zip(rows, dem):
    d = re.sub(r"^void dmb::", "", d.split("(")[0])
    if flt and flt not in d:
        continue
    print("%-90s %5s %5s %5s %7s %7s %8s %7s %4s" % (d[:90], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("TotalSGPRs", r.get("SGPRs", "?")),
          r.get("SGPRs Spill", "?"), r.get("VGPRs Spill", "?"), r.get("ScratchSize [bytes/lane]", "?"),
          r.get("LDS Size [bytes/block]", "?"), r.get("Occupancy [waves/SIMD]", "?")))
