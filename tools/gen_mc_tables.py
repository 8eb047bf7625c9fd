#!/usr/bin/env python3
"""Generate the marching-cubes tables used by the meshing engine.

Upstream ITMMeshingEngine uses the classic 256-entry edge / triangle tables (ITMMeshingEngine.h
buildVertList + edgeTable + triangleTable) with this cube numbering:

    vertices  0:(0,0,0) 1:(1,0,0) 2:(1,1,0) 3:(0,1,0) 4:(0,0,1) 5:(1,0,1) 6:(1,1,1) 7:(0,1,1)
    edges     0:0-1 1:1-2 2:2-3 3:3-0 4:4-5 5:5-6 6:6-7 7:7-4 8:0-4 9:1-5 10:2-6 11:3-7
    cubeIndex bit k set  <=>  sdf(vertex k) < 0

The upstream tables themselves are not in /root/reference (empty submodule) nor anywhere on this
machine, so the tables are GENERATED here from the topology instead of being copied:

  * edgeTable[c]  = the edges whose end points differ in sign (this part is forced);
  * triTable[c]   = on every cube face the crossed edges are joined by segments — on an ambiguous
    face (4 crossings) always so that each NEGATIVE corner is cut off on its own, a rule that
    depends on the face's signs only, hence is the same for the two cubes sharing the face
    (=> watertight) —, the segments are chained into closed polygons, each polygon is wound so
    that its normal points to the negative side (the convention of the classic table: case 1 is
    the triangle 0,8,3; ITMMesh::WriteOBJ then lists the faces reversed) and fan-triangulated
    from its lowest edge.

The script checks: every row uses exactly the edges of edgeTable; <= 5 triangles per row; and on
random sign fields the union of all cubes' triangles is a closed, consistently oriented surface
(every interior mesh edge is used exactly twice, in opposite directions).

Usage: python tools/gen_mc_tables.py [out.h ...]   (default: both in-tree copies)
"""
import itertools
import os
import random
import sys

V = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
E = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
EDGE_OF = {}
for i, (a, b) in enumerate(E):
    EDGE_OF[(a, b)] = EDGE_OF[(b, a)] = i
# faces as cycles of 4 vertices, counter-clockwise seen from outside the cube (checked below)
FACES = [(0, 3, 2, 1), (4, 5, 6, 7), (0, 1, 5, 4), (3, 7, 6, 2), (0, 4, 7, 3), (1, 2, 6, 5)]


def sub(a, b): return tuple(x - y for x, y in zip(a, b))
def add(a, b): return tuple(x + y for x, y in zip(a, b))
def dot(a, b): return sum(x * y for x, y in zip(a, b))
def cross(a, b): return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])
def mid(e): return tuple((V[E[e][0]][k] + V[E[e][1]][k]) / 2.0 for k in range(3))


def row(c):
    neg = [(c >> k) & 1 for k in range(8)]
    crossed = [e for e, (a, b) in enumerate(E) if neg[a] != neg[b]]
    succ = {}
    for f in FACES:
        # walk the face counter-clockwise (seen from outside).  Every maximal run of negative corners
        # is cut off by one segment, directed from the edge where the walk LEAVES the run (neg -> pos)
        # to the edge where it ENTERED it (pos -> neg): the polygon normals then point to the
        # negative side, as in the classic table (case 1 = 0,8,3).  Two separate runs (the ambiguous
        # face) give two segments, each cutting off one negative corner.
        for i in range(4):
            if neg[f[i]] and not neg[f[(i + 1) % 4]]:          # leaves a run between corners i and i+1
                j = i
                while neg[f[(j - 1) % 4]]:
                    j -= 1                                       # first corner of the run
                src = EDGE_OF[(f[i], f[(i + 1) % 4])]
                dst = EDGE_OF[(f[(j - 1) % 4], f[j % 4])]
                assert src not in succ
                succ[src] = dst
    assert sorted(succ) == sorted(crossed) and sorted(succ.values()) == sorted(crossed), (c, succ)
    loops, seen = [], set()
    for start in sorted(crossed):
        if start in seen:
            continue
        loop, cur = [], start
        while cur not in seen:
            seen.add(cur); loop.append(cur)
            cur = succ[cur]
        assert cur == start
        loops.append(loop)
    tris = []
    for loop in loops:
        assert len(loop) >= 3
        for i in range(1, len(loop) - 1):
            tris.append((loop[0], loop[i], loop[i + 1]))
    assert len(tris) <= 5, (c, tris)
    used = sorted(set(itertools.chain.from_iterable(tris)))
    assert used == sorted(crossed), (c, used, crossed)
    return sum(1 << e for e in crossed), tris, loops


def tables():
    for f, n in zip(FACES, [(0, 0, -1), (0, 0, 1), (0, -1, 0), (0, 1, 0), (-1, 0, 0), (1, 0, 0)]):
        assert cross(sub(V[f[1]], V[f[0]]), sub(V[f[2]], V[f[1]])) == n
    et, tt, ll = [], [], []
    for c in range(256):
        e, t, l = row(c)
        et.append(e); tt.append(t); ll.append(l)
    assert tt[1] == [(0, 8, 3)] and tt[2] == [(0, 1, 9)] and tt[254] == [(0, 3, 8)]
    return et, tt, ll


def check_watertight(tt, ll, n=7, trials=40, seed=7):
    """On random sign fields: (1) the polygons of neighbouring cubes meet along every cube face in
    opposite directions (each directed boundary segment once, its reverse once) => closed and
    consistently oriented; (2) every triangle row is a triangulation of its polygons (the directed
    triangle edges that are not polygon sides cancel inside the cube)."""
    for c in range(256):
        sides = {}
        for loop in ll[c]:
            for i in range(len(loop)):
                sides[(loop[i], loop[(i + 1) % len(loop)])] = 1
        cnt = {}
        for tri in tt[c]:
            for i in range(3):
                cnt[(tri[i], tri[(i + 1) % 3])] = cnt.get((tri[i], tri[(i + 1) % 3]), 0) + 1
        for (a, b), k in cnt.items():
            assert k == 1
            assert ((a, b) in sides) != ((b, a) in cnt), (c, a, b)
        assert all(s_ in cnt for s_ in sides)
    rng = random.Random(seed)
    for _ in range(trials):
        sign = {(x, y, z): rng.random() < 0.45 for x in range(n + 1) for y in range(n + 1) for z in range(n + 1)}
        for k in list(sign):  # positive shell => the surface is closed inside the grid
            if 0 in k or n in k:
                sign[k] = False
        count = {}
        for x, y, z in itertools.product(range(n), repeat=3):
            c = sum(1 << k for k in range(8) if sign[(x + V[k][0], y + V[k][1], z + V[k][2])])
            for loop in ll[c]:
                g = []
                for e in loop:
                    a = add((x, y, z), V[E[e][0]]); b = add((x, y, z), V[E[e][1]])
                    g.append((min(a, b), max(a, b)))
                for i in range(len(g)):
                    key = (g[i], g[(i + 1) % len(g)])
                    count[key] = count.get(key, 0) + 1
        for (a, b), k in count.items():
            assert k == 1 and count.get((b, a), 0) == 1, "polygon side not matched by the neighbouring cube"


def emit(path, et, tt):
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_mc_tables.py — do not edit.  Marching-cubes tables in the cube numbering of\n"
                "// upstream's ITMMeshingEngine.h (bit k of the index = sdf(vertex k) < 0); see the generator for the\n"
                "// construction rule and the checks it runs (edge sets, <= 5 triangles, watertightness).\n"
                "#pragma once\n\n#ifndef MC_TABLE_ATTR\n#define MC_TABLE_ATTR static const\n#endif\n\n")
        f.write("MC_TABLE_ATTR int kMcEdgeTable[256] = {\n")
        for i in range(0, 256, 8):
            f.write("    " + ", ".join("0x%03x" % e for e in et[i:i + 8]) + ",\n")
        f.write("};\n\n// 16 entries per case: up to 5 triangles (edge indices), -1 terminated\n")
        f.write("MC_TABLE_ATTR signed char kMcTriTable[256][16] = {\n")
        for c in range(256):
            flat = list(itertools.chain.from_iterable(tt[c]))
            flat += [-1] * (16 - len(flat))
            f.write("    {" + ", ".join("%2d" % v for v in flat) + "},\n")
        f.write("};\n")


if __name__ == "__main__":
    et, tt, ll = tables()
    check_watertight(tt, ll)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = sys.argv[1:] or [os.path.join(root, "dynslam_amd", "csrc", "mc_tables.h"), os.path.join(root, "oracle", "mc_tables.h")]
    for o in outs:
        emit(o, et, tt)
    print("rows with 1..5 triangles:", [sum(1 for t in tt if len(t) == k) for k in range(1, 6)])
