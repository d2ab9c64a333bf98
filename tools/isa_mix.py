"""Static instruction mix of the innermost loops of a gfx950 kernel (hipcc -S output): how many MFMA / VALU / LDS /
VMEM / SALU instructions one trip of each loop issues, and the issue cycles they need on one SIMD (MFMA 32x32x16 f16:
8 passes = 32 cycles; a wave64 VALU op: 4 cycles on the 16-lane SIMD, 8 for the packed / transcendental / 64-bit ones
counted as 4 here -- a LOWER bound; ds_* and global_* one issue cycle each, their data paths are separate).  A loop is
a label that a later s_cbranch jumps back to.
usage: python tools/isa_mix.py file.s <kernel-name-substring> [...]"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("ds_read", "ds_load")):
        return "lds_read"
    if op.startswith(("ds_write", "ds_store")):
        return "lds_write"
    if op.startswith(("global_load", "buffer_load", "scratch_load", "flat_load")):
        return "vmem_load"
    if op.startswith(("global_store", "buffer_store", "scratch_store", "flat_store", "global_atomic")):
        return "vmem_store"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernels(path):
    cur, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, body = m.group(1), []
            continue
        if cur is not None:
            if ".end_amdhsa_kernel" in line or line.startswith("\t.section"):
                yield cur, body
                cur = None
            else:
                body.append(line.rstrip("\n"))


def loops(body):
    labels = {}
    ins = []
    for line in body:
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        t = line.strip()
        if not t or t.startswith((";", ".")):
            continue
        ins.append(t.split(";")[0].strip())
    out = []
    for k, t in enumerate(ins):
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", t)
        if m and m.group(1) in labels and labels[m.group(1)] <= k:
            out.append((labels[m.group(1)], k))
    # innermost only
    inner = [l for l in out if not any(o != l and l[0] <= o[0] and o[1] <= l[1] for o in out)]
    return ins, inner


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    for name, body in kernels(path):
        if pats and not any(p in name for p in pats):
            continue
        ins, inner = loops(body)
        print("%s: %d instructions, %d innermost loop(s)" % (name, len(ins), len(inner)))
        for a, b in inner:
            c = Counter(classify(t.split()[0]) for t in ins[a:b + 1])
            if c["mfma"] == 0 and b - a < 40:
                continue
            cyc_mfma, cyc_valu = 32 * c["mfma"], 4 * c["valu"]
            other = c["lds_read"] + c["lds_write"] + c["vmem_load"] + c["vmem_store"] + c["salu"] + c["waitcnt"]
            print("  loop of %4d instr: mfma %3d  valu %4d  lds_read %3d  lds_write %3d  vmem_load %3d  vmem_store %2d  "
                  "salu %3d  waitcnt %3d  barrier %d | issue cycles: mfma %5d, valu >= %5d, other >= %4d -> valu/mfma %.2f"
                  % (b - a + 1, c["mfma"], c["valu"], c["lds_read"], c["lds_write"], c["vmem_load"], c["vmem_store"],
                     c["salu"], c["waitcnt"], c["barrier"], cyc_mfma, cyc_valu, other, cyc_valu / max(1, cyc_mfma)))


if __name__ == "__main__":
    main()
