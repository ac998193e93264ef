"""Print a window of a rocprofv3 kernel trace (CSV) as a timeline: kernel, start and duration in us relative to the first row.
usage: python tools/trace_window.py <kernel_trace.csv> [first_row] [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
a = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:a + n]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-48s q%-3s start %8.2f dur %7.2f" % (r["Kernel_Name"][:48], r.get("Queue_Id", "?"), (s - t0) / 1e3, (e - s) / 1e3))
