"""rocprofv3 passes of scripts/profile_train.sh (gpurun_out/prof_train_<tag>/) -> profiles/<tag>_train_kernel_stats.csv (per-kernel
time of the traced run: calls, total, mean, share) and profiles/<tag>_train_pmc.csv: per kernel and STEP the time, the matrix
arithmetic it executed (SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 FLOP), the HBM bytes it moved (2 x FETCH_SIZE + WRITE_SIZE KiB -- gfx950's
FETCH_SIZE counts half of the fetched bytes, scripts/summarize_profiles.py) and, from those, the fraction of the FP32 matrix peak
(157.3 TFLOP/s) and of the HBM peak (8 TB/s) it ran at: the larger of the two is the kernel's roofline fraction, `bound` says which.
Durations come from the trace pass (counter passes stretch launches); counters are summed over the launches of ONE step."""
import collections
import csv
import glob
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_train_" + tag)
DST = os.path.join(ROOT, "profiles")
STEPS_TRACED, STEPS_PMC = 8, 3      # train_bench: warm-up + timed steps + the phase-split step (2 + 5 + 1, 1 + 1 + 1)
PEAK_TF, PEAK_GBS = 157.3, 8000.0


def short(name):
    n = name.replace("void ", "").replace("dmb::", "")
    return n.split("(")[0][:120]


def find(d, pat):
    p = glob.glob(os.path.join(SRC, d, "**", pat), recursive=True)
    return p[0] if p else None


dur = collections.defaultdict(list)
for r in csv.DictReader(open(find("trace", "*kernel_trace.csv"))):
    dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
nsteps = None
log = open(os.path.join(SRC, "trace.log")).read()
total = sum(sum(v) for v in dur.values())
rows = sorted(dur.items(), key=lambda kv: -sum(kv[1]))
with open(os.path.join(DST, tag + "_train_kernel_stats.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python scripts/train_bench.py --steps 5 --warmup 2  (MI355X; PSMNet cost path, training mode, "
            "batch 4 x 256x512 crops, max_disp 192, Adam; scripts/profile_train.sh)\n")
    f.write("kernel,calls,total_ms,avg_us,percent,min_us,max_us\n")
    for k, v in rows:
        if sum(v) / total < 0.0005:
            continue
        f.write("%s,%d,%.3f,%.1f,%.2f,%.1f,%.1f\n" % (k, len(v), sum(v) / 1e3, sum(v) / len(v), 100.0 * sum(v) / total, min(v), max(v)))

cnt = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(int)
for d in ("pmc_fetch", "pmc_write", "pmc_sq"):
    p = find(d, "*counter_collection.csv")
    if not p:
        continue
    seen = set()
    for r in csv.DictReader(open(p)):
        k = short(r["Kernel_Name"])
        cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if d == "pmc_sq" and r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            launches[k] += 1
with open(os.path.join(DST, tag + "_train_pmc.csv"), "w") as f:
    f.write("# per training STEP (counter passes: %d steps, trace: %d): ms = mean kernel time per step from the trace pass; mfma_gflop = "
            "SQ_INSTS_VALU_MFMA_MOPS_F32 * 512 / 1e9; hbm_gb = (2 * FETCH_SIZE + WRITE_SIZE) KiB; frac_mfma = mfma_gflop / ms / %.1f TFLOP/s; "
            "frac_hbm = hbm_gb / ms / %.0f GB/s; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)\n" % (STEPS_PMC, STEPS_TRACED, PEAK_TF, PEAK_GBS))
    f.write("kernel,launches_per_step,ms_per_step,share,mfma_gflop,hbm_gb,frac_mfma,frac_hbm,bound,mfma_util\n")
    tot_ms = tot_gf = 0.0
    for k, v in rows:
        ms = sum(v) / 1e3 / STEPS_TRACED
        if ms < 0.01:
            continue
        c = cnt.get(k, {})
        gf = c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512 / 1e9 / STEPS_PMC
        gb = (2 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024 / 1e9 / STEPS_PMC
        fm, fh = gf / ms / PEAK_TF, gb / ms * 1e3 / PEAK_GBS
        util = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (c["GRBM_GUI_ACTIVE"] / 8 * 1024) if c.get("GRBM_GUI_ACTIVE") else float("nan")
        tot_ms += ms
        tot_gf += gf
        f.write("%s,%.1f,%.3f,%.2f,%.2f,%.3f,%.3f,%.3f,%s,%.3f\n" % (k, len(v) / STEPS_TRACED, ms, 100.0 * sum(v) / total, gf, gb, fm, fh,
                                                                    "mfma" if fm >= fh else "hbm", util))
    f.write("# sum of the rows: %.2f ms of kernel time per step, %.1f GFLOP of matrix arithmetic -> %.3f of the FP32 matrix peak over the kernel time\n"
            % (tot_ms, tot_gf, tot_gf / tot_ms / PEAK_TF))
print(open(os.path.join(DST, tag + "_train_pmc.csv")).read()[:6000])
print(log[-600:])
