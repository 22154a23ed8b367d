"""Per training step of a rocprofv3 --kernel-trace CSV: span, busy time, idle time inside the step and the idle time in front of it
(a large one = the host synchronised there), and the step's gaps above 5 us with the kernel they precede.
usage: python tools/step_gaps.py gpurun_out/<dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "deform_frame_bwd_kernel" in r["Kernel_Name"] or "deform_head_wgrad_tr_all_kernel" in r["Kernel_Name"]]   # a step's last launch
prev_end_t = None
for n in range(1, len(ends)):
    sel = rows[ends[n - 1] + 1:ends[n] + 1]
    t0 = int(sel[0]["Start_Timestamp"]); pe = t0; busy = 0; gaps = []
    for r in sel:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s - pe > 5000: gaps.append("%.0f us before %s" % ((s - pe) / 1e3, r["Kernel_Name"].split("(")[0][-40:]))
        pe = max(pe, e); busy += e - s
    before = (t0 - prev_end_t) / 1e3 if prev_end_t else 0.0
    prev_end_t = pe
    print("step %3d: idle before %9.1f us  span %7.1f  busy %7.1f  idle inside %6.1f  launches %d  %s"
          % (n, before, (pe - t0) / 1e3, busy / 1e3, (pe - t0 - busy) / 1e3, len(sel), "; ".join(gaps)))
