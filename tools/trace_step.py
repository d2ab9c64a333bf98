"""Print the launch sequence of the LAST step out of a rocprofv3 kernel trace csv.
usage: python tools/trace_step.py <kernel_trace.csv> <first-kernel-substring> [out.txt|-] [nth-from-end]
The step is taken to start at the last launch whose name contains the substring (nth-from-end = 2: the step before the
last one, printed up to the start of the last -- bench.py's final call is the 15-image repeated-job check)."""
import csv
import re
import sys

path, marker = sys.argv[1], sys.argv[2]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                     r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", "")))
rows.sort()
starts = [i for i, r in enumerate(rows) if marker in r[2]]
# one step = from the last marker run start to the end (markers may repeat inside a step: take the
# first of the last contiguous group)
nth = int(sys.argv[4]) if len(sys.argv) > 4 else 1
i0, i1 = starts[-1], len(rows)
for _ in range(nth):
    if _:
        i1, i0 = i0, max(j for j in starts if j < i0)
    while i0 - 1 in starts:
        i0 -= 1
out = open(sys.argv[3], "w") if len(sys.argv) > 3 and sys.argv[3] != "-" else sys.stdout
t0 = rows[i0][0]
busy = 0
for s, e, name, grid, wg in rows[i0:i1]:
    short = re.sub(r"^void ", "", name)
    short = re.sub(r"\(.*$", "", short).replace("disn::", "")
    busy += e - s
    out.write("%9.1f us  +%8.1f us  grid %-9s %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, grid, short))
out.write("span %.1f us, kernel busy %.1f us, launches %d\n" % ((rows[i1 - 1][1] - t0) / 1e3, busy / 1e3, i1 - i0))
