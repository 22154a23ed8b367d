"""Prints the top kernels of a rocprofv3 --kernel-trace --stats CSV directory and (optionally) writes a markdown summary."""
import csv, glob, sys
d = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else None
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
lines = ["| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
for r in rows[:24]:
    lines.append("| %s | %s | %.3f | %.1f | %s |" % (r["Name"][:90].replace("|", "/"), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
txt = "\n".join(lines)
print(txt)
if out:
    open(out, "w").write(txt + "\n")
