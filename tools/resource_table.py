"""Per-kernel resource table (VGPRs, SGPRs, scratch, LDS, waves per SIMD) of every HIP source, from the compiler's own
kernel-resource-usage remarks.  usage: python tools/resource_table.py profiles/r02_kernel_resources.md"""
import os, re, subprocess, sys
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "e-d3dgs_amd", "csrc")
rows = []
for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
    flags = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-munsafe-fp-atomics", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
    if src == "preprocess.hip":
        flags.insert(0, "-ffp-contract=off")
    out = subprocess.run(["/opt/rocm/bin/hipcc"] + flags, cwd=CSRC, capture_output=True, text=True).stderr
    cur = None
    for ln in out.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = {"file": src, "name": m.group(1)}
            rows.append(cur)
            continue
        for key, pat in (("vgpr", r"VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, ln)
            if m and cur is not None and " VGPRs Spill" not in ln and "SGPRs Spill" not in ln:
                cur.setdefault(key, int(m.group(1)))
def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except Exception:
        return n
lines = ["# Kernel resources (gfx950, `-O3`, the compiler's kernel-resource-usage remarks; static LDS only -- dynamic LDS is set at launch)", "",
         "| file | kernel | VGPRs | SGPRs | scratch B/lane | static LDS B | waves/SIMD |", "|---|---|---|---|---|---|---|"]
for r in rows:
    if "vgpr" not in r or "rocprim" in r["name"] or "hipcub" in r["name"]:
        continue
    nm = demangle(r["name"])
    nm = re.sub(r"\(.*", "", nm).replace("void ", "").replace("ed3::", "")
    lines.append("| %s | `%s` | %d | %d | %d | %d | %d |" % (r["file"], nm[:80], r["vgpr"], r.get("sgpr", 0), r.get("scratch", 0), r.get("lds", 0), r.get("occ", 0)))
open(sys.argv[1], "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:6]), "\n...", len(lines) - 4, "kernels")
