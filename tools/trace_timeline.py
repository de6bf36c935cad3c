"""usage: python tools/trace_timeline.py <rocprofv3 kernel_trace.csv> [anchor-kernel-substring]
Prints the launches of the LAST full step of the trace (from the last-but-one launch of the anchor kernel to the last one): start, idle gap
before the launch, duration, grid size, name.  What docs/measurement_log.md quotes as "the timeline of a call"."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "solve_generic2"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
i0, i1 = idx[-2], idx[-1]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  grid {r['Grid_Size_X']:>8s} {r['Kernel_Name'][:70]}")
    busy += e - s
    prev_end = e
print(f"busy {busy / 1e3:.1f} us of {(prev_end - t0) / 1e3:.1f} us")
