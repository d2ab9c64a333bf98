"""Average duration per (kernel, grid size) of a rocprofv3 kernel-trace csv.  usage: trace_summary.py trace.csv [substr]"""
import csv, re, sys
from collections import defaultdict
acc = defaultdict(list)
order = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"])).replace("disn::", "")
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        k = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", ""))
        if k not in acc:
            order.append(k)
        acc[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in order:
    v = sorted(acc[k])
    print("%-46s grid %-8s wg %-4s n %4d  median %8.2f us  min %8.2f  max %8.2f" % (
        k[0][:46], k[1], k[2], len(v), v[len(v) // 2] / 1e3, v[0] / 1e3, v[-1] / 1e3))
