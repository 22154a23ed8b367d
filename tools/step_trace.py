"""Prints the kernel launches of ONE forward+backward step from a rocprofv3 --kernel-trace CSV: offset, gap to the previous kernel's
end, duration, name.  usage: python tools/step_trace.py gpurun_out/<dir> [step index from the end, default 2]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# a forward+backward step: from the deform_frag_kernel before a render_forward to the deform_frame_bwd_kernel after the render_backward
ends = [i for i, r in enumerate(rows) if "deform_frame_bwd_kernel" in r["Kernel_Name"] or "deform_head_wgrad_tr_all_kernel" in r["Kernel_Name"]]   # a step's last launch
e = ends[-back]
prev = ends[-back - 1]
sel = rows[prev + 1:e + 1]
t0 = int(sel[0]["Start_Timestamp"]); prev_end = t0; tot = 0
for r in sel:
    s, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  +%6.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (en - s) / 1e3, r["Kernel_Name"][:110]))
    prev_end = max(prev_end, en); tot += en - s
print("launches", len(sel), "busy %.1f us" % (tot / 1e3), "span %.1f us" % ((prev_end - t0) / 1e3))
