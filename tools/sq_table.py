"""profiles/<tag>_pmc_sq.md from one rocprofv3 --pmc SQ pass and the counter-free kernel stats of the same command.
usage: python tools/sq_table.py gpurun_out/pmc_<tag>_sq gpurun_out/prof_<tag>_stats profiles/<tag>_pmc_sq.md <tag>"""
import collections
import csv
import glob
import sys

pmc_dir, stats_dir, out, tag = sys.argv[1:5]
f = glob.glob(pmc_dir + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
ks = glob.glob(stats_dir + "/**/*kernel_stats.csv", recursive=True)
if ks:
    for r in csv.DictReader(open(ks[0])):
        dur[r["Name"]] = float(r["AverageNs"]) / 1e3
cols = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
        "SQ_LDS_BANK_CONFLICT"]
lines = [f"# {tag} SQ counter pass (rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT "
         "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES)", "",
         "Command: `rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --no-cpu-baseline --no-other-modes [--train-only from round 4 on: "
         "every launch a training-step launch] --steps 4 --warmup 1` (C3, default multiply mode). Means per dispatch.",
         f"`MFMA busy` = SQ_VALU_MFMA_BUSY_CYCLES / (average duration x 2.4 GHz x 1024 SIMDs), duration from the counter-free "
         f"`{tag}_bench_c3_kernel_stats.md` (all launches of the kernel in that run: for the deformation forward that mixes the keeping "
         "launches of training with the non-keeping ones of the fps pass);",
         "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles (x4 for cycles).", "",
         "| kernel | n | MFMA busy cycles | wave cycles (x4) | wait any (x4) | wait LDS (x4) | VALU active (x4) | LDS active (x4) | LDS conflict | avg us | MFMA busy |",
         "|---|---|---|---|---|---|---|---|---|---|---|"]
rows = []
for k, d in agg.items():
    if not ("ed3::" in k):
        continue
    n = len(next(iter(d.values())))
    m = {c: sum(d[c]) / len(d[c]) if c in d else 0.0 for c in cols}
    us = dur.get(k)
    busy = "%.0f %%" % (100 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (us * 1e-6 * 2.4e9 * 1024)) if us and m["SQ_VALU_MFMA_BUSY_CYCLES"] > 0 else "-"
    rows.append((us or 0.0, "| %s | %d | %s | %s | %s |" % (k[:72], n, " | ".join("%.3g" % m[c] for c in cols), ("%.1f" % us) if us else "-", busy)))
for _, ln in sorted(rows, reverse=True):
    lines.append(ln)
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[8:]))
