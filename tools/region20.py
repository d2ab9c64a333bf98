"""Where the driver's 20-step timed region goes: rocprofv3 kernel trace of `bench.py --steps 20 --warmup 5 --no-extras`;
prints, for the last N calls (a call starts with resize_kernel), start of its first kernel, end of its last
mlp_fused_kernel<false...> / final_dot_kernel, relative to the first of them.
usage: python tools/region20.py <kernel_trace.csv> [ncalls=4]"""
import csv, re, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                     re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("disn::", "").replace("void ", ""), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
starts = [i for i, r in enumerate(rows) if r[2].startswith("resize_kernel<1>")][-n:]
t0 = rows[starts[0]][0]
ends = [r for r in rows if r[0] >= t0 and ("mlp_fused_kernel<false" in r[2] or "final_dot" in r[2])]
for i in starts:
    print("call of %d images starts at %8.1f us" % (rows[i][3] // (224 * 224 * 3 // 4 * 4) if False else rows[i][3] // 131072, (rows[i][0] - t0) / 1e3))
for r in ends:
    print("   %-40s grid %8d ends at %8.1f us" % (r[2][:40], r[3], (r[1] - t0) / 1e3))
reg = [r for r in rows if r[0] >= t0]
cur_s, cur_e, busy = reg[0][0], reg[0][1], 0
for s, e, _, _ in reg[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("span %.1f us, GPU busy (union) %.1f us" % ((max(r[1] for r in reg) - t0) / 1e3, busy / 1e3))
