"""Build libdisn_amd.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m disn_amd.csrc.build [--force] [--verbose] [--tuning]

The library is built IN-TREE (disn_amd/csrc/libdisn_amd.so) so that it travels to the GPU
box with the repo snapshot.  elementwise.hip is compiled with -ffp-contract=off (bit-exact
rows A/D/E/F/J); everything else with the default contraction.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libdisn_amd.so")
LIB_TUNING = os.path.join(HERE, "libdisn_amd_tuning.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

SOURCES = {
    "gemm_mfma.hip": [],
    "gemv.hip": [],
    "gemm_tn_mfma.hip": [],
    "gemm_bf16_mfma.hip": [],
    "backward.hip": [],
    "backward_img.hip": ["-munsafe-fp-atomics", "-ffp-contract=off"],  # same lerp weights as the forward
    "train.hip": [],
    "cam_head.hip": [],
    "mlp_small.hip": [],
    "mlp_fused.hip": [],
    "conv_h2.hip": ["-mllvm", "-pragma-unroll-threshold=1000000"],
    "conv_h2w.hip": ["-mllvm", "-pragma-unroll-threshold=1000000"],   # the unrolled chunk bodies exceed the default 16 k
    "dense_h2.hip": [],
    "dense_h2w.hip": [],
    "elementwise.hip": ["-ffp-contract=off"],
    "marching_cubes.hip": ["-ffp-contract=off"],
    "api.hip": [],
    "host_util.cpp": ["-msse4.2"],
}
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
HEADERS = ["kernels.hpp", "tuning.hpp", "h2_common.hpp", "mc_tables.h", os.path.join(ROOT, "include", "disn_amd.h")]


def _digest(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build(force: bool = False, verbose: bool = False, tuning: bool = False) -> str:
    """tuning=True: the same sources with -DDISN_TUNING (run-time knobs + disn_tuning_set, csrc/tuning.hpp)
    as libdisn_amd_tuning.so -- for tools/ only; the product library has no knobs."""
    lib = LIB_TUNING if tuning else LIB
    extra = ["-DDISN_TUNING"] if tuning else []
    srcs = {s: f + extra for s, f in SOURCES.items() if os.path.exists(os.path.join(HERE, s))}
    hdrs = [h if os.path.isabs(h) else os.path.join(HERE, h) for h in HEADERS]
    objs, jobs = [], []
    for src, flags in srcs.items():
        sp = os.path.join(HERE, src)
        tag = _digest([sp] + hdrs, " ".join(COMMON + flags))
        obj = os.path.join(HERE, "build", "%s.%s.o" % (src, tag))
        objs.append(obj)
        if force or not os.path.exists(obj):
            jobs.append((sp, obj, flags))
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)

    def compile_one(job):
        sp, obj, flags = job
        cmd = [HIPCC] + COMMON + flags + ["-c", sp, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (sp, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(compile_one, jobs))
    stamp = os.path.join(HERE, "build", "link_tuning.stamp" if tuning else "link.stamp")
    want = _digest(objs) if all(os.path.exists(o) for o in objs) else ""
    have = open(stamp).read().strip() if os.path.exists(stamp) else ""
    if force or jobs or not os.path.exists(lib) or want != have:
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        with open(stamp, "w") as f:
            f.write(want)
    # objects of earlier source states (other digests) would otherwise pile up and travel with every snapshot;
    # the twin build's objects (product <-> tuning) are recognised by ITS stamp's digest list and kept
    keep = set(objs)
    other = os.path.join(HERE, "build", "objs_tuning.txt" if not tuning else "objs.txt")
    if os.path.exists(other):
        keep.update(l.strip() for l in open(other))
    with open(os.path.join(HERE, "build", "objs.txt" if not tuning else "objs_tuning.txt"), "w") as f:
        f.write("\n".join(objs))
    for fn in os.listdir(os.path.join(HERE, "build")):
        fp = os.path.join(HERE, "build", fn)
        if fn.endswith(".o") and fp not in keep:
            os.remove(fp)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv,
                tuning="--tuning" in sys.argv))
