"""Timeline of a rocprofv3 *_kernel_trace.csv: start (ms since the first dispatch), duration, queue, kernel,
grid -- to see what overlaps what.  Usage: python tools/trace_timeline.py <kernel_trace.csv> [min_start_ms]"""
import csv
import re
import sys


def short(name):
    m = re.search(r"(k_[a-z0-9_]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name.split("(")[0][:40]


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
print("start_ms,dur_ms,queue,kernel,grid,wg")
for r in rows:
    s = (int(r["Start_Timestamp"]) - t0) * 1e-6
    if s < lo:
        continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    print(f"{s:.3f},{d:.3f},{r.get('Queue_Id', '?')},\"{short(r['Kernel_Name'])}\",{r.get('Grid_Size_X') or r.get('Grid_Size')},{r.get('Workgroup_Size_X') or r.get('Workgroup_Size')}")
