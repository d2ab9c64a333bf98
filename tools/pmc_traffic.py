"""profiles/pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
tools/prof_kernels.py (PROF_STEPS=1): HBM-side bytes of the conv family of ONE step (every conv GEMM
launch + the split-K reduce / stream-K fix-up that directly follows it) and of the stand-alone
gathers.  Corrections as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE (KB) x2 for wide
coalesced reads; WRITE_SIZE calibrated on the gather at N = 262144, whose output bytes are known.
usage: python tools/pmc_traffic.py <dir with pmc_FETCH_SIZE.csv pmc_WRITE_SIZE.csv> > profiles/pmc_traffic.json"""
import collections
import csv
import json
import os
import sys

d = sys.argv[1]


def load(name, counter):
    out = collections.OrderedDict()
    for r in csv.DictReader(open(os.path.join(d, name))):
        if r["Counter_Name"] != counter:
            continue
        k = int(r["Dispatch_Id"])
        e = out.setdefault(k, {"name": r["Kernel_Name"], "grid": r["Grid_Size"], "v": 0.0})
        e["v"] += float(r["Counter_Value"])
    return out


fetch = load("pmc_FETCH_SIZE.csv", "FETCH_SIZE")
write = load("pmc_WRITE_SIZE.csv", "WRITE_SIZE")
ids = sorted(fetch)


def is_conv(n):
    return ("conv_h2_kernel" in n or "conv_h2w_kernel" in n or "conv1_1_direct_kernel" in n or ("gemm_bf16_mfma" in n and ", 1, " in n)
            or ("gemm_f32_mfma" in n and (", 1, 0>" in n or ", 2, 0>" in n)))


# the first step = from the first conv GEMM to the 13th
conv, taken, count = [], set(), 0
for pos, i in enumerate(ids):
    if is_conv(fetch[i]["name"]) and count < 13:
        conv.append(i)
        count += 1
        if pos + 1 < len(ids):
            nxt = fetch[ids[pos + 1]]["name"]
            if "splitk_reduce" in nxt or "streamk_fixup" in nxt:
                conv.append(ids[pos + 1])
# the LAST 13 conv launches, when there are more than the steps' (prof_kernels.py ends with one PROF_BATCH-image stack)
all_conv = [i for i in ids if is_conv(fetch[i]["name"])]
batched = all_conv[-13:] if len(all_conv) >= 39 else []
batch_images = int(os.environ.get("PROF_BATCH", "16"))
gathers = [i for i in ids if "gather_kernel" in fetch[i]["name"] and "project" not in fetch[i]["name"]]
g_small = [i for i in gathers if int(fetch[i]["grid"]) < 1_000_000][-1]
g_big = [i for i in gathers if int(fetch[i]["grid"]) >= 1_000_000][-1]
cal = 262144 * 5888 / 1024.0 / write[g_big]["v"]
g_taps = [i for i in ids if "project_gather_taps_kernel" in fetch[i]["name"]]
g_wave = [i for i in ids if "project_gather_taps_wave_kernel" in fetch[i]["name"]]   # round 6: calls of >= 10 240 points
g_fold = [i for i in ids if "gather_fold_kernel" in fetch[i]["name"]]
fused = [i for i in ids if "mlp_fused_kernel" in fetch[i]["name"]]
build = os.environ.get("PMC_BUILD", "")
try:
    build = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "disn_amd", "csrc", "build",
                              "BUILD_ID")).read().strip() or build
except Exception:
    pass


def hbm(idl):
    f = sum(fetch[i]["v"] for i in idl)
    w = sum(write[i]["v"] for i in idl if i in write)
    return {"launches": len(idl), "fetch_kb": f, "fetch_corrected_kb": 2 * f, "write_kb": w,
            "hbm_bytes": (2 * f + cal * w) * 1024.0}


out = {"build": build, "counter_files": os.environ.get("PMC_BUILD", ""), "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/prof_kernels.py workload (PROF_STEPS=1)",
       "note": "KB as reported; gfx950: FETCH_SIZE x2 for wide coalesced reads; WRITE_SIZE calibrated on the gather",
       "conv_family_per_step": dict(hbm(conv), kernels=sorted({fetch[i]["name"].split("(")[0] for i in conv}),
                                    algorithmic_bytes_note="conv_h2 build: weights 59 MB (two f16 planes, 4 B/weight) + layer "
                                    "inputs ~36 MB + outputs ~54 MB (+ 14 MB pooled copies); every workgroup re-reads its "
                                    "n-block's weights and its halo through L2 (hits there are not counted, MALL hits "
                                    "are); three-term build: weights 88 MB + split-K partials"),
       "conv_family_batched": (dict(hbm(batched), images=batch_images,
                                    note="one disn_vgg16_conv_stack call on %d images (what bench.py's main line "
                                         "submits per call): the weights are fetched once for the batch" % batch_images)
                               if batched else None),
       "gather_n2048": dict(hbm([g_small]), algorithmic_bytes=2048 * 29440),
       "gather_n262144": dict(hbm([g_big]), algorithmic_bytes=262144 * 29440),
       "gather_from_taps_n2048": (dict(hbm(g_taps[-1:]), algorithmic_bytes=2048 * (16 * 5888 + 5888),
                                       note="disn_encode_query: 16 tap reads + 1 write per output float4; the taps "
                                            "(24.5 MB) are L2 / MALL resident, so the memory-side bytes are far "
                                            "below the algorithmic reads") if g_taps else None),
       "gather_taps_n2048": (dict(hbm(g_taps[-1:]), algorithmic_bytes=2048 * 29440) if g_taps else None),
       "gather_fold_n65536": (dict(hbm(g_fold[-1:]), algorithmic_bytes=65536 * 12288) if g_fold else None),
       "mlp_fused_n65536": ({"global": hbm([i for i in fused if "<false" in fetch[i]["name"] or "Lb0" in fetch[i]["name"]][-1:]),
                             "local": hbm([i for i in fused if "<true" in fetch[i]["name"] or "Lb1" in fetch[i]["name"]][-1:]),
                             "note": "per 65536 points: weights re-streamed per 128-point tile stay in L2; the local "
                                     "stream adds 8 KB/point of pmap rows (L2 / MALL resident)"} if fused else None),
       "small_set_b16": (lambda fe, gl, ga: {
           "gather_split": dict(hbm((g_wave or ga)[-1:]), algorithmic_bytes=16 * 2048 * 29440,
                                kernel="project_gather_taps_wave_kernel" if g_wave else "project_gather_taps_kernel",
                                note="16 x 2048 points, split rows; algorithmic = the 29 440 B/point convention of roofline_gather "
                                     "(the one-wave-per-point kernel requests ~36 KB/point of L2 / MALL-resident tap pixels and writes 6144 B/point)"),
           "local_feat": dict(hbm(fe[-1:]), algorithmic_bytes=16 * 2048 * 6144 + 5308416,
                              note="split rows read once + one pass over the 5.3 MB weight image (every workgroup re-reads it from L2)"),
           "global": hbm([i for i in gl if i > fe[-1]][:1])} if fe else None)(
           [i for i in fused if "true, false, true" in fetch[i]["name"] or "Lb1ELb0ELb1" in fetch[i]["name"]],
           [i for i in fused if ("<false" in fetch[i]["name"] or "Lb0" in fetch[i]["name"]) and int(fetch[i]["grid"]) == 65536],
           [i for i in g_taps if int(fetch[i]["grid"]) >= 4000000]),
       "write_calibration": {"factor": cal, "basis": "gather_kernel at N=262144 writes exactly 262144*5888 B"}}
print(json.dumps(out, indent=1))
