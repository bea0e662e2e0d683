"""Tables of objects for lt_submit_objects: the arrays of several polygon objects laid end to end, and
per object the matrix the host built for it and its range of the shared vertex array.

The reference draws a world one object at a time -- M_render_polygon_object (EngPoly.cpp:499-525)
takes fewer than 1024 vertices and at most 256 polygons (Engine.h:432-433) -- so a frame of many
objects is many calls; one table is one GPU batch with the same frame as the calls in turn."""
from typing import List, Sequence, Tuple

import numpy as np

OBJECT_INTS = 16


class ObjectTable:
    """objects int32[nobj][16], polyobj int32[np], verts, polys, polyaux (or None), polyverts, max_edges"""

    def __init__(self, objects, polyobj, verts, polys, polyaux, polyverts, max_edges):
        self.objects, self.polyobj, self.verts, self.polys = objects, polyobj, verts, polys
        self.polyaux, self.polyverts, self.max_edges = polyaux, polyverts, max_edges


def build_table(objs: Sequence[Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray, np.ndarray]]) -> ObjectTable:
    """objs: (matrix int32[12], verts int32[nv][4], polys int32[np][4], polyverts int32[npv][6], polyaux int32[np][4] or None)
    per object, in draw order.  polys[:, 3] (first record of each polygon) is moved to the object's place in the
    joined record array; the records' vertex indices stay relative to their object."""
    table = np.zeros((len(objs), OBJECT_INTS), np.int32)
    vs, ps, pvs, auxs, owner = [], [], [], [], []
    vbase = 0
    pvbase = 0
    with_aux = any(o[4] is not None for o in objs)
    for i, (m, verts, polys, polyverts, polyaux) in enumerate(objs):
        verts = np.ascontiguousarray(verts, np.int32).reshape(-1, 4)
        polys = np.ascontiguousarray(polys, np.int32).reshape(-1, 4).copy()
        polyverts = np.ascontiguousarray(polyverts, np.int32).reshape(-1, 6)
        table[i, :12] = np.asarray(m, np.int32).reshape(12)
        table[i, 12] = vbase
        table[i, 13] = verts.shape[0]
        polys[:, 3] += pvbase
        vs.append(verts); ps.append(polys); pvs.append(polyverts)
        owner.append(np.full(polys.shape[0], i, np.int32))
        if with_aux:
            auxs.append(np.zeros((polys.shape[0], 4), np.int32) if polyaux is None else np.ascontiguousarray(polyaux, np.int32).reshape(-1, 4))
        vbase += verts.shape[0]
        pvbase += polyverts.shape[0]
    polys = np.concatenate(ps) if ps else np.zeros((0, 4), np.int32)
    me = int(polys[:, 2].max()) if polys.shape[0] else 0
    return ObjectTable(table, np.concatenate(owner) if owner else np.zeros(0, np.int32),
                       np.concatenate(vs) if vs else np.zeros((0, 4), np.int32), polys,
                       np.concatenate(auxs) if with_aux else None,
                       np.concatenate(pvs) if pvs else np.zeros((0, 6), np.int32), min(max(me, 1), 15))


def split_soup(verts: np.ndarray, polys: np.ndarray, polyverts: np.ndarray, per_object: int, polyaux=None) -> List[tuple]:
    """A triangle soup (every triangle its own three vertices and four records, as raytracing_amd.scenes makes
    them) cut into objects of `per_object` triangles: [(verts, polys, polyverts, polyaux)] with indices and first
    records counted from each object's own arrays."""
    out = []
    n = polys.shape[0]
    for a in range(0, n, per_object):
        b = min(a + per_object, n)
        p = polys[a:b].copy()
        f0 = int(p[0, 3])
        f1 = int(p[-1, 3] + p[-1, 2] + 1)
        pv = polyverts[f0:f1].copy()
        v0, v1 = int(pv[:, 0].min()), int(pv[:, 0].max()) + 1
        pv[:, 0] -= v0
        p[:, 3] -= f0
        out.append((verts[v0:v1].copy(), p, pv, None if polyaux is None else polyaux[a:b].copy()))
    return out
