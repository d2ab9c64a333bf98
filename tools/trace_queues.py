"""Compact multi-queue timeline of the last ~N launches of a rocprofv3 kernel trace: start, duration, queue, name."""
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"),
                 re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"])).replace("disn::", "")))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 90
rows = rows[-n:]
t0 = rows[0][0]
for s, e, q, name in rows:
    print("%9.1f +%6.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, name[:40]))
