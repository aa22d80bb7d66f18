"""Prints the last N kernels of a rocprofv3 --kernel-trace CSV as a timeline (start offset, duration, queue, name):
python tools/trace_tail.py <kernel_trace.csv> [N=200] [skip_last=0]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = rows[len(rows) - n - skip:len(rows) - skip]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    nm = r["Kernel_Name"].split("(")[0].replace("void bessx::", "").replace("bessx::", "").split("<")[0]
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f %7.1f  q%-3s %s" % ((a - t0) / 1e3, (b - a) / 1e3, r.get("Queue_Id", "?"), nm))
