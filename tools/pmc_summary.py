"""Merges the FETCH_SIZE and WRITE_SIZE passes of tools/pmc_pass.sh into a markdown table and profiles/<tag>_pmc_summary.json.
usage: python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_fetch_write.md profiles/r01_pmc_summary.json
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE tallies the 128-B requests of a wide
coalesced read at 64 B (MI355X_MICROARCH.md, HBM / rocprofv3 section); WRITE_SIZE is exact for 16-B-per-lane stores and
float atomics.  The x2 is calibrated for 16 B/lane streams only (gathers of 64-B records may be over-corrected)."""
import collections, csv, glob, json, re, sys

def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg

fd, wd, md, js = sys.argv[1:5]
cmd = sys.argv[5] if len(sys.argv) > 5 else "python bench.py --no-cpu-baseline --steps 4 --warmup 1"
# optional: a bench line of the same build (JSON file) whose kernels[*].algorithmic_bytes_per_launch give the ratio column
alg = {}
if len(sys.argv) > 6:
    try:
        for k, v in json.load(open(sys.argv[6])).get("kernels", {}).items():
            if isinstance(v, dict) and v.get("algorithmic_bytes_per_launch"):
                alg[k.split(" ")[0]] = v["algorithmic_bytes_per_launch"]
    except Exception:
        alg = {}
F, Wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
rows = []
for k in F:
    f = sum(F[k]) / len(F[k]); w = sum(Wr.get(k, [0])) / max(len(Wr.get(k, [0])), 1)
    rows.append((k, len(F[k]), f, w, (2 * f + w) * 1024))
rows.sort(key=lambda r: -r[4])
lines = ["# PMC passes (rocprofv3 --kernel-trace --pmc, separate passes for FETCH_SIZE and WRITE_SIZE)", "",
         "Command: `rocprofv3 --kernel-trace --pmc <CTR> -- %s` (C3 workload)." % cmd,
         "Units: KB per dispatch (mean over dispatches). Per MI355X_MICROARCH.md, on gfx950 FETCH_SIZE reports half of a wide",
         "coalesced stream, so HBM bytes ~= (2*FETCH_SIZE + WRITE_SIZE)*1024; the x2 is calibrated for 16 B/lane streams only.", "",
         "| kernel | dispatches | FETCH_SIZE KB | WRITE_SIZE KB | (2F+W) MB | algorithmic MB | traffic / algorithmic |", "|---|---|---|---|---|---|---|"]
out, ratio = {}, {}
for k, n, f, w, b in rows:
    if b < 1e5:
        continue
    m = re.search(r"ed3::(\w+)(<[^>(]*>)?", k)
    name = (m.group(1) + (m.group(2) or "").replace(" ", "")) if m else None
    a = alg.get(name)
    lines.append("| %s | %d | %.0f | %.0f | %.1f | %s | %s |" % (k[:70].replace("|", "/"), n, f, w, b / 1e6, "%.1f" % (a / 1e6) if a else "", "%.2f" % (b / a) if a else ""))
    if name:   # keyed by kernel name incl. template arguments, e.g. "deform_forward_b3_kernel<4,3>"
        out.setdefault(name, b)
        if a:
            ratio.setdefault(name, b / a)
open(md, "w").write("\n".join(lines) + "\n")
json.dump({"workload": "C3", "command": cmd, "traffic_over_algorithmic": ratio, "algorithmic_bytes_per_launch": alg, "source": md + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
           "hbm_bytes_per_launch": out,
           "note": "FETCH_SIZE doubled per the gfx950 correction for wide coalesced reads; gathers of 64-B records may be over-corrected"},
          open(js, "w"), indent=1)
print("\n".join(lines[6:26]))
