"""Summarise a gpurun_out/<tag>/ rocprofv3 directory (tools/gpu_prof.sh) into text:
per-kernel duration stats and per-dispatch PMC counters (FETCH_SIZE/WRITE_SIZE in KB as
reported; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE under-reports wide coalesced streaming
reads by 2x -- the 'fetch_x2' column applies that correction; WRITE_SIZE is uncalibrated).
usage: python tools/pmc_summary.py gpurun_out/r01p > profiles/r01_pmc_summary.txt"""
import collections, csv, os, sys

d = sys.argv[1]


def agg(f):
    out = collections.OrderedDict()
    p = os.path.join(d, f)
    if not os.path.exists(p):
        return out
    for r in csv.DictReader(open(p)):
        key = (int(r['Dispatch_Id']), r['Kernel_Name'].replace('disn::', '').split('(')[0][:44], r['Grid_Size'])
        e = out.setdefault(key, {})
        e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        e['_ns'] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    return out


files = {k: agg(k) for k in os.listdir(d) if k.startswith('pmc_') and k.endswith('.csv')}
fetch = files.get('pmc_FETCH_SIZE.csv', {})
write = files.get('pmc_WRITE_SIZE.csv', {})
sq = files.get('pmc_SQ_WAVE_CYCLES_SQ_WAIT_ANY_SQ_WAIT_INST_ANY_SQ_ACTIVE_INST_ANY.csv', {})
mf = files.get('pmc_SQ_VALU_MFMA_BUSY_CYCLES_SQ_BUSY_CYCLES_GRBM_GUI_ACTIVE.csv', {})
lds = files.get('pmc_SQ_LDS_BANK_CONFLICT_SQ_LDS_IDX_ACTIVE.csv', {})
print("# per-dispatch counters (workload: tools/prof_kernels.py; one counter set per pass)")
print("%-44s %9s %9s | %11s %11s %11s | %6s %6s %6s | %9s %7s" % (
    "kernel", "grid", "dur_us", "fetch_KB", "fetch_x2_KB", "write_KB", "wait%", "stall%", "act%", "mfma_busy", "ldsconf%"))
for k in fetch:
    if not any(s in k[1] for s in ("gemm", "gather", "gemv", "resize", "splitk", "maxpool")):
        continue
    f = fetch[k]; w = write.get(k, {}); s = sq.get(k, {}); m = mf.get(k, {}); l = lds.get(k, {})
    wc = s.get('SQ_WAVE_CYCLES', 0) or 1
    print("%-44s %9s %9.1f | %11.1f %11.1f %11.1f | %6.1f %6.1f %6.1f | %9.3g %7.2f" % (
        k[1], k[2], f['_ns'] / 1e3, f.get('FETCH_SIZE', 0), 2 * f.get('FETCH_SIZE', 0), w.get('WRITE_SIZE', 0),
        100 * s.get('SQ_WAIT_ANY', 0) / wc, 100 * s.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * s.get('SQ_ACTIVE_INST_ANY', 0) / wc,
        m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0), 100 * l.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, l.get('SQ_LDS_IDX_ACTIVE', 1))))
for name in ("bench_kernel_stats.csv", "kernels_kernel_stats.csv"):
    p = os.path.join(d, name)
    if os.path.exists(p):
        print("\n# %s (rocprofv3 --kernel-trace --stats)" % name)
        for r in csv.DictReader(open(p)):
            if float(r['Percentage']) < 0.3:
                continue
            print("%-60s calls %5s avg_us %9.2f total_ms %8.3f %5s%%" % (
                r['Name'].replace('disn::', '')[:60], r['Calls'], float(r['AverageNs']) / 1e3,
                float(r['TotalDurationNs']) / 1e6, r['Percentage']))
