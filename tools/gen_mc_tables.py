"""Derive the marching-cubes case table (no table is copied from anywhere).

Corner / edge numbering (the usual one): corners 0..7 at (0,0,0) (1,0,0) (1,1,0) (0,1,0) (0,0,1)
(1,0,1) (1,1,1) (0,1,1); edges 0..11 = 0-1 1-2 2-3 3-0 4-5 5-6 6-7 7-4 0-4 1-5 2-6 3-7.
Case index: bit i set <=> corner i is INSIDE (value < iso).

For every case: the cut edges are paired face by face; a face with four cut edges (diagonal
corners inside) is ambiguous and is resolved by ONE rule that looks only at that face's corner
signs -- each inside corner is cut off on its own -- so two cells sharing the face always agree and
the surface has no cracks (the original Lorensen-Cline table with complement symmetry does not
have that property).  The pairs form closed loops (every cut edge lies on two faces); each loop is
oriented so that its normal points from inside to outside and is fan-triangulated.

    python tools/gen_mc_tables.py          # writes disn_amd/csrc/mc_tables.h

``build_tables()`` is also imported by the oracle / tests.
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import numpy as np

CORNERS = np.array([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)], float)
EDGES = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
FACES = [(0, 1, 2, 3), (4, 5, 6, 7), (0, 1, 5, 4), (3, 2, 6, 7), (0, 3, 7, 4), (1, 2, 6, 5)]
EDGE_OF = {frozenset(e): i for i, e in enumerate(EDGES)}


def case_triangles(mask: int) -> List[Tuple[int, int, int]]:
    inside = [(mask >> i) & 1 for i in range(8)]
    adj: Dict[int, List[int]] = {}

    def link(a, b):
        adj.setdefault(a, []).append(b)
        adj.setdefault(b, []).append(a)

    for f in FACES:
        es = [EDGE_OF[frozenset((f[k], f[(k + 1) % 4]))] for k in range(4)]       # edge k joins corner k, k+1
        cut = [k for k in range(4) if inside[f[k]] != inside[f[(k + 1) % 4]]]
        if len(cut) == 2:
            link(es[cut[0]], es[cut[1]])
        elif len(cut) == 4:
            for k in range(4):                 # cut off every inside corner: edges k-1 and k meet at corner k
                if inside[f[k]]:
                    link(es[(k - 1) % 4], es[k])
    tris: List[Tuple[int, int, int]] = []
    seen = set()
    for start in sorted(adj):
        if start in seen:
            continue
        loop, prev, cur = [start], None, start
        seen.add(start)
        while True:
            nxt = [n for n in adj[cur] if n != prev] if prev is not None else [adj[cur][0]]
            if prev is not None and adj[cur][0] == adj[cur][1]:      # 2-cycle cannot happen on a cube
                raise AssertionError
            n = nxt[0]
            if n == start:
                break
            loop.append(n)
            seen.add(n)
            prev, cur = cur, n
        pts = np.array([(CORNERS[EDGES[e][0]] + CORNERS[EDGES[e][1]]) / 2 for e in loop])
        normal = np.zeros(3)
        for i in range(len(pts)):                                     # Newell
            p, q = pts[i], pts[(i + 1) % len(pts)]
            normal += np.cross(p, q)
        ins = {c for e in loop for c in EDGES[e] if inside[c]}
        outs = {c for e in loop for c in EDGES[e] if not inside[c]}
        direction = CORNERS[list(outs)].mean(0) - CORNERS[list(ins)].mean(0)
        if np.dot(normal, direction) < 0:
            loop = loop[::-1]
        # fan apex: prefer a rotation none of whose diagonals joins two cut edges lying on one cube
        # face -- such a diagonal lies IN that face and can coincide with a segment or diagonal of the
        # neighbouring cell (a non-manifold edge).  All rotations give a crack-free surface.
        def in_face(a, b):
            return any(set(EDGES[a]) | set(EDGES[b]) <= set(f) for f in FACES)
        best = 0
        for r in range(len(loop)):
            rot = loop[r:] + loop[:r]
            if not any(in_face(rot[0], rot[i]) for i in range(2, len(rot) - 1)):
                best = r
                break
        loop = loop[best:] + loop[:best]
        for i in range(1, len(loop) - 1):
            tris.append((loop[0], loop[i], loop[i + 1]))
    return tris


def build_tables():
    """-> (ntri[256] int32, tri[256][3*MAXT] int8 padded with -1, MAXT)"""
    cases = [case_triangles(m) for m in range(256)]
    maxt = max(len(c) for c in cases)
    ntri = np.array([len(c) for c in cases], np.int32)
    tri = -np.ones((256, 3 * maxt), np.int8)
    for m, c in enumerate(cases):
        flat = [e for t in c for e in t]
        tri[m, :len(flat)] = flat
    return ntri, tri, maxt


def edge_geometry():
    """per cube edge: lower corner offset (dx,dy,dz) and axis -> the grid edge it lies on"""
    out = []
    for a, b in EDGES:
        lo = np.minimum(CORNERS[a], CORNERS[b]).astype(int)
        axis = int(np.nonzero(CORNERS[a] != CORNERS[b])[0][0])
        out.append((int(lo[0]), int(lo[1]), int(lo[2]), axis))
    return out


def write_header(path: str) -> None:
    ntri, tri, maxt = build_tables()
    eg = edge_geometry()
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_mc_tables.py -- do not edit.  Marching-cubes case table derived by\n"
                "// face-consistent loop tracing (see the generator); case bit i = corner i inside (value < iso).\n"
                "#pragma once\n"
                "#ifndef DISN_MC_QUAL\n#define DISN_MC_QUAL static const\n#endif\n")
        f.write("#define DISN_MC_MAXT %d\n" % maxt)
        f.write("DISN_MC_QUAL unsigned char kMcNtri[256] = {%s};\n" % ",".join(str(int(v)) for v in ntri))
        f.write("DISN_MC_QUAL signed char kMcTri[256][%d] = {\n" % (3 * maxt))
        for m in range(256):
            f.write("  {%s},\n" % ",".join(str(int(v)) for v in tri[m]))
        f.write("};\n")
        f.write("// cube edge -> (dx,dy,dz) of its lower grid point and its axis\n")
        f.write("DISN_MC_QUAL unsigned char kMcEdge[12][4] = {%s};\n" % ",".join("{%d,%d,%d,%d}" % e for e in eg))


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "disn_amd", "csrc", "mc_tables.h")
    write_header(out)
    ntri, tri, maxt = build_tables()
    print("wrote", out, "max triangles per cell", maxt, "histogram", np.bincount(ntri).tolist())
