"""The timed region of a short bench.py run out of a rocprofv3 kernel trace csv: everything from the N-th-last launch whose
name contains <marker> (default: the first kernel of the region's first call) to the last kernel -- span, union of busy
time, idle gaps, and the start / end of every call's first and last kernel.
usage: python tools/trace_region.py <kernel_trace.csv> <marker> <calls in the region>"""
import csv, re, sys
path, marker, ncalls = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("disn::", "").replace("void ", "")))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
i0 = marks[-ncalls]
t0 = rows[i0][0]
reg = rows[i0:]
print("region: %d launches, span %.1f us" % (len(reg), (max(r[1] for r in reg) - t0) / 1e3))
# union of busy intervals and gaps
cur_s, cur_e, busy, gaps = reg[0][0], reg[0][1], 0, []
for s, e, n in reg[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((cur_e, s))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("GPU busy (union) %.1f us; idle inside the region %.1f us in %d gaps" % (busy / 1e3, sum(b - a for a, b in gaps) / 1e3, len(gaps)))
for a, b in sorted(gaps, key=lambda g: g[0] - g[1])[:8]:
    print("   gap %.1f us at %.1f us" % ((b - a) / 1e3, (a - t0) / 1e3))
for i in marks[-ncalls:]:
    print("call starts (%s) at %.1f us" % (marker, (rows[i][0] - t0) / 1e3))
fd = [r for r in reg if "final_dot" in r[2]]
for r in fd:
    print("final_dot ends at %.1f us" % ((r[1] - t0) / 1e3))
# concurrency profile: time with >= 2 kernels running
ev = sorted([(s, 1) for s, e, n in reg] + [(e, -1) for s, e, n in reg])
lvl, last, t2 = 0, ev[0][0], 0
for t, d in ev:
    if lvl >= 2:
        t2 += t - last
    lvl += d
    last = t
print("time with >= 2 kernels in flight: %.1f us" % (t2 / 1e3))
