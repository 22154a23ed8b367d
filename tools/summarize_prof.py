"""Prints the top kernels of a rocprofv3 --kernel-trace --stats output directory (CSV, or the sqlite .db when no CSV was
written) and (optionally) writes a markdown summary."""
import csv, glob, sqlite3, sys
d = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else None
lines = ["| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
dbs = sorted(glob.glob(d + "/**/*.db", recursive=True))
csvs = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))
if csvs and not (dbs and len(sys.argv) > 3 and sys.argv[3] == "db"):
    rows = list(csv.DictReader(open(csvs[0])))
    for r in rows[:24]:
        lines.append("| %s | %s | %.3f | %.1f | %s |" % (r["Name"][:90].replace("|", "/"), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
else:
    c = sqlite3.connect(dbs[-1])
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = list(c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    for r in rows[:24]:
        lines.append("| %s | %d | %.3f | %.1f | %.2f |" % (r[0][:90].replace("|", "/"), r[1], r[2] / 1e6, r[3] / 1e3, 100.0 * r[2] / tot))
txt = "\n".join(lines)
print(txt)
if out:
    open(out, "w").write(txt + "\n")
