"""MFMA utilisation per kernel from one rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE pass over
tools/prof_mfma.py.  SQ_VALU_MFMA_BUSY_CYCLES sums, over all 1024 SIMDs, the cycles their matrix pipe was busy (32 per
v_mfma_f32_32x32x16_f16: MI355X_MICROARCH.md; checked: conv2_2 of eight images = 2 709 504 MFMAs x 32 = the counter to
the digit); GRBM_GUI_ACTIVE is reported SUMMED OVER THE 8 XCDs (17 "GHz" against the dispatch's duration), so one
XCD's active shader cycles are GRBM_GUI_ACTIVE / 8.
utilisation = busy / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); effective clock = GRBM_GUI_ACTIVE / 8 / duration.
usage: python tools/pmc_mfma.py <counter_collection.csv> > profiles/r03x_pmc_mfma.txt"""
import collections, csv, sys

rows = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = int(r["Dispatch_Id"])
    e = rows.setdefault(k, {"name": r["Kernel_Name"].replace("disn::", "").split("(")[0], "grid": int(r["Grid_Size"]),
                            "wg": int(r.get("Workgroup_Size", 0) or 0),
                            "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
last = collections.OrderedDict()
for k, e in rows.items():
    if "MFMA" not in "".join(e.keys()):
        continue
    if not any(s in e["name"] for s in ("conv_h2", "dense_h2", "conv1_1", "mlp_fused")):
        continue
    last[(e["name"], e["grid"])] = e          # the last dispatch of each (kernel, grid)
print("# MFMA utilisation, last dispatch of each (kernel, grid); counters of ONE pass (profiled clocks are ~5 % below un-profiled)")
print("%-52s %9s %9s %14s %12s %7s %6s" % ("kernel", "grid", "us", "mfma busy cyc", "gui active", "util", "GHz"))
tot = {}
for (name, grid), e in last.items():
    busy, gui = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), e.get("GRBM_GUI_ACTIVE", 0.0)
    util = busy / (128.0 * gui) if gui else 0.0
    print("%-52s %9d %9.1f %14.0f %12.0f %6.1f%% %6.2f" % (name[:52], grid, e["ns"] / 1e3, busy, gui, 100 * util,
                                                          gui / 8.0 / e["ns"] if e["ns"] else 0))
    fam = None          # (families are not summed: the same (kernel, grid) serves several layers; see the rows)
    if fam:
        t = tot.setdefault(fam, [0.0, 0.0, 0.0])
        t[0] += busy; t[1] += gui; t[2] += e["ns"]
print("# per family (sums over the rows above)")
for fam, (busy, gui, ns) in tot.items():
    print("%-28s busy %14.0f  gui active %12.0f  duration %8.1f us  utilisation %5.1f%%" % (fam, busy, gui, ns / 1e3,
                                                                                         100 * busy / (128 * gui) if gui else 0))
