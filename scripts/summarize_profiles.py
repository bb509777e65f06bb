"""Turn the rocprofv3 CSVs of scripts/profile.sh (under gpurun_out/prof_<tag>/) into the small tracked files under
profiles/: per-kernel time stats, per-kernel PMC means, and pmc_dominant.json (HBM bytes per launch of the dominant
kernel, which bench.py reports as roofline.traffic).

Counter handling follows MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
counts 64 B per 128-B request, i.e. HALF the bytes actually fetched -- calibrated here on soft_argmin_kernel, whose
one pass over the [4,192,544,960] FP32 volume must fetch 1.604 GB -- so fetch bytes = 2 * FETCH_SIZE * 1024;
WRITE_SIZE is exact (trilinear_kernel writes exactly 1.604 GB)."""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
# the dominant kernel's instantiation and its launch shape: the 48 x 4 row pairs at 544x960, the 32 x 4 row pairs at 384x1248 (KITTI)
DOM = os.environ.get("DOM_PREFIX", "conv3d_s1_kernel<S1Cfg<0, 32, 4, 48")
DOM_SHAPE = [int(v) for v in os.environ.get("DOM_SHAPE", "4,32,48,136,240").split(",")]
WHAT = " ".join(([("--config " + os.environ["PROF_CONFIG"])] if os.environ.get("PROF_CONFIG") else []) +
                ([("--batch " + os.environ["PROF_BATCH"])] if os.environ.get("PROF_BATCH") else []))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
DST = os.path.join(ROOT, "profiles")
os.makedirs(DST, exist_ok=True)


def short(name):
    n = name.replace("void ", "").replace("dmb::", "")
    return n.split("(")[0]


# Per-kernel statistics from the per-dispatch trace.  Under rocprofv3 a launch now and then comes back 5-10x its normal duration
# (seen at ~0.9 s into a traced run, the following launches slow for a few ms: 15-28 ms for a 2.4 ms kernel; 300 back-to-back steps
# WITHOUT the profiler show nothing of the kind, scripts/attic/step_jitter_probe.py: max 27.23 against a median of 26.93 ms): such a
# launch (> 4x the median of its kernel) is left out of the averages and listed in the last column.
tr = os.path.join(SRC, "trace", "trace_kernel_trace.csv")
d = collections.defaultdict(list)
for r in csv.DictReader(open(tr)):
    d[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
stats = []
for k, v in d.items():
    med = sorted(v)[len(v) // 2]
    keep = [t for t in v if t <= 4 * med]
    out = [t for t in v if t > 4 * med]
    stats.append((k, keep, out))
total = sum(sum(keep) for _, keep, _ in stats)
stats.sort(key=lambda e: -sum(e[1]))
with open(os.path.join(DST, tag + "_kernel_stats.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extras %s--steps 5 --warmup 2  (MI355X; scripts/profile.sh)\n" % (WHAT + " " if WHAT else ""))
    f.write("# from the per-dispatch trace; a launch stretched to more than 4x its kernel's median by the profiler is left out and listed in the last column (scripts/summarize_profiles.py)\n")
    f.write("kernel,calls,total_ms,avg_us,percent,min_us,max_us,left_out\n")
    for k, keep, out in stats:
        if sum(keep) / total < 0.0005:
            continue
        f.write("%s,%d,%.3f,%.1f,%.2f,%.1f,%.1f,%s\n" % (k, len(keep), sum(keep) / 1e3, sum(keep) / len(keep), 100.0 * sum(keep) / total,
                                                        min(keep), max(keep), " ".join("%.0fus" % t for t in out)))

# The any-Ci row-pair stride-1 kernel serves both the 32->32 layers (6 launches per step) and the 64->32 layer (1 per
# step) under ONE kernel name: split its dispatches by duration (the 64->32 launch does twice the arithmetic) so that the
# dominant 32->32 launches can be compared with bench.py's live HIP-event figure.
with open(os.path.join(DST, tag + "_kernel_stats.csv"), "a") as f:
    f.write("# split of the shared stride-1 kernel by launch duration (short = Ci 32, long = Ci 64)\n")
    for k, keep, out in stats:
        if not k.startswith(DOM):
            continue
        lo = min(keep)
        for name, sel in (("Ci=32", [t for t in keep if t < 1.5 * lo]), ("Ci=64", [t for t in keep if t >= 1.5 * lo])):
            if sel:
                f.write("%s [%s],%d,%.3f,%.1f,,%.1f,%.1f,\n" % (k, name, len(sel), sum(sel) / 1e3, sum(sel) / len(sel), min(sel), max(sel)))

pmc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
import glob
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_fetch_gwc", "pmc_write_gwc", "pmc_sq_gwc"):
    paths = glob.glob(os.path.join(SRC, d, "**", "*counter_collection.csv"), recursive=True)
    if not paths:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))   # dispatch -> counter -> sum over its rows
    meta = {}
    for r in csv.DictReader(open(paths[0])):
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        meta[r["Dispatch_Id"]] = (short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    # the shared stride-1 kernel serves Ci = 32 and Ci = 64 launches under one name: a launch that takes >= 1.5x the shortest
    # one of its name in this pass is a Ci = 64 launch and gets its own row, so that a row is ONE layer shape
    shortest = collections.defaultdict(lambda: float("inf"))
    for did in per:
        k, us = meta[did]
        shortest[k] = min(shortest[k], us)
    for did, cs in per.items():
        k, us = meta[did]
        if d.endswith("_gwc") and not k.startswith("gwc"):
            continue
        if k.startswith(DOM) and us >= 1.5 * shortest[k]:
            k += " [Ci=64]"
        for c, v in cs.items():
            pmc[k][c].append(v)
            if d.startswith("pmc_sq") and c == "GRBM_GUI_ACTIVE":
                dur[k].append(us)
counters = ["FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_BUSY_CYCLES",
            "SQ_WAVE_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"]
with open(os.path.join(DST, tag + "_pmc.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc <one group per pass> -- python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 (gwc_mfma_kernel: scripts/kcase.py gwc); "
            "means per launch.  hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 correction, see "
            "scripts/summarize_profiles.py); mfma_util = MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs); "
            "mfma_gflop = MOPS_F32 * 512 / 1e9\n")
    f.write("kernel,launches," + ",".join(counters) + ",hbm_bytes,mfma_util,mfma_gflop,pmc_pass_us\n")
    for k, v in sorted(pmc.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
        if not k.startswith(("conv3d", "deconv3d", "trilinear", "soft_argmin", "volume", "epe", "conf_head", "gwc", "conv2d", "catconv", "copy_window")):
            continue
        m = {c: (sum(v[c]) / len(v[c]) if v.get(c) else float("nan")) for c in counters}
        hbm = 2 * m["FETCH_SIZE"] * 1024 + m["WRITE_SIZE"] * 1024
        util = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024) if m["GRBM_GUI_ACTIVE"] == m["GRBM_GUI_ACTIVE"] else float("nan")
        f.write("%s,%d,%s,%.4g,%.4f,%.2f,%.1f\n" % (k, len(v.get("GRBM_GUI_ACTIVE", v.get("FETCH_SIZE", []))),
                                                   ",".join("%.6g" % m[c] for c in counters), hbm, util,
                                                   m["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512 / 1e9,
                                                   sum(dur[k]) / len(dur[k]) if dur[k] else float("nan")))
# pmc_dominant.json is a VIEW of the dominant kernel's row of <tag>_pmc.csv (same launches, same means): bench.py reads it
dom = [k for k in pmc if k.startswith(DOM) and not k.endswith("[Ci=64]")]
dom.sort(key=lambda k: -sum(pmc[k].get("GRBM_GUI_ACTIVE", [0])))
if dom:
    v = pmc[dom[0]]
    fetch, write = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]), sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
    json.dump({"kernel": dom[0], "round": tag, "derived_from": "profiles/%s_pmc.csv" % tag, "launches": len(v["FETCH_SIZE"]), "shape": DOM_SHAPE,
               "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
               "hbm_bytes_per_launch": 2 * fetch * 1024 + write * 1024,
               "note": "gfx950: FETCH_SIZE counts half of the fetched bytes (calibrated on soft_argmin_kernel); WRITE_SIZE exact"},
              open(os.path.join(DST, "pmc_dominant.json" if not WHAT else "pmc_dominant_%s.json" % tag), "w"), indent=1)
print(open(os.path.join(DST, tag + "_kernel_stats.csv")).read())
print(open(os.path.join(DST, tag + "_pmc.csv")).read()[:3000])
