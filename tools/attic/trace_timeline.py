"""Timeline of the last kernels in a rocprofv3 --kernel-trace CSV: start / end relative to the first one
shown, per queue.  usage: trace_timeline.py <dir or kernel_trace.csv> [how many]"""
import csv, glob, os, sys
path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"q{r.get('Queue_Id', '?'):>3} {a/1e3:9.1f} -> {b/1e3:9.1f} us ({(b-a)/1e3:7.1f})  {r['Kernel_Name'][:70]}")
