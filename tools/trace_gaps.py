"""Kernel timeline of a rocprofv3 --kernel-trace run: per kernel start/duration/gap to the previous kernel's end.
usage: python tools/trace_gaps.py <dir with *kernel_trace.csv> [last N kernels]"""
import csv, glob, sys
d = sys.argv[1]; last = int(sys.argv[2]) if len(sys.argv) > 2 else 80
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
prev_end = None; out = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append((r["Kernel_Name"][:60], (s - int(rows[0]["Start_Timestamp"])) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0.0))
    prev_end = e
if len(sys.argv) > 3:      # third argument: print only the span between the first and last kernel whose name contains it
    hit = [i for i, o in enumerate(out) if sys.argv[3] in o[0]]
    out = out[max(hit[0] - 3, 0): hit[-1] + 6] if hit else []
for name, t, dur, gap in out[-last:]:
    print(f"{t:12.1f} us  dur {dur:8.1f}  gap {gap:8.1f}  {name}")
