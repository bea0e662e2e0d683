"""Screen-clipped lines drawn after filled polygons.  G_line clips with dimension 2
(GraphicsBase.cpp:191-193), so a clipped end point's depth is a leftover of the polygon clips
before it in the clippers' static stores (Clipp2D.cpp:99-104).  With HW_track_clipper_stores
the host mirror makes the reference's clipper calls for every polygon and clips lines on the
host; the frames must equal the ones the reference build drew
(tests/golden/stale_depth_kats.json).  Without tracking a clipped end point takes depth 0."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

from raytracing_amd import scenes as S

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as MG  # noqa: E402

GOLD = json.load(open(os.path.join(HERE, "golden", "stale_depth_kats.json")))
W, H = MG.WIRE_W, MG.WIRE_H


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def eng():
    import torch  # noqa: F401
    from raytracing_amd import _lib
    _lib.lib()
    L = C.CDLL(os.path.join(os.path.dirname(_lib.SO_PATH), "liblt_engine.so"))
    L.lte_cbuf.restype = C.POINTER(C.c_int32)
    L.lte_zbuf.restype = C.POINTER(C.c_int32)
    L.lte_init(W, H)
    yield L
    L.lte_track_clipper_stores(0)
    L.lte_init_clipping(0, 0, W - 1, H - 1)


def _run(L, steps, whole_objects):
    keep = []
    for st in steps:
        if st[0] == "clear":
            L.lte_clear(st[1], 255, 255, 255)
        elif st[0] == "gpolys":
            for (e, t, col) in st[2]:
                tt = np.ascontiguousarray(t, dtype=np.int32)
                keep.append(tt)
                L.lte_g_polygon(st[1], _p(tt), e, col, 0, 0, 0, 0)
        elif st[0] == "object":
            _, sc, rt, obj = st
            if sc.textures:
                tex = np.ascontiguousarray(sc.textures[0][0], dtype=np.int32)
                keep.append(tex)
                L.lte_set_texture(0, _p(tex), sc.textures[0][1])
            v, po, pa, pv = sc.verts.copy(), sc.polys.copy(), sc.polyaux.copy(), sc.polyverts.copy()
            keep += [v, po, pa, pv]
            if whole_objects:          # M_render_polygon_object
                L.lte_render_object(_p(v), v.shape[0], _p(po), _p(pa), _p(pv), po.shape[0], rt, 0, 0, 0, 0, 0, 0, sc.log_focus,
                                    *obj[3:], *obj[:3])
            else:                      # the transform, then M_render_polygon per polygon
                L.lte_render_polys_one_by_one(_p(v), v.shape[0], _p(po), _p(pa), _p(pv), po.shape[0], rt, sc.log_focus,
                                              *obj[3:], *obj[:3])
        else:
            for l in st[1]:
                a, b = np.ascontiguousarray(l[0:3]), np.ascontiguousarray(l[3:6])
                L.lte_g_line(_p(a), _p(b), int(l[6]), 255, 255, 255)
    L.lte_flush()
    c = np.ctypeslib.as_array(L.lte_cbuf(), shape=(H, W)).copy()
    z = np.ctypeslib.as_array(L.lte_zbuf(), shape=(H, W)).copy()
    return c, z


@pytest.mark.parametrize("whole_objects", [False, True])
@pytest.mark.parametrize("name", sorted(GOLD))
def test_tracked_stores_match_the_reference_frames(eng, name, whole_objects):
    eng.lte_track_clipper_stores(1)
    c, z = _run(eng, MG.stale_depth_cases()[name], whole_objects)
    assert S.frame_digest(z) == GOLD[name]["depth"] and S.frame_digest(c) == GOLD[name]["colour"]


@pytest.mark.parametrize("name", sorted(GOLD, reverse=True))
def test_frames_do_not_depend_on_the_case_before(eng, name):
    """the stores are process-wide; every case overwrites all of them first, so any order of cases gives
    the reference's frames (here the reverse of the order above)"""
    eng.lte_track_clipper_stores(1)
    c, z = _run(eng, MG.stale_depth_cases()[name], name < "n")
    assert S.frame_digest(z) == GOLD[name]["depth"] and S.frame_digest(c) == GOLD[name]["colour"]


def test_untracked_lines_take_depth_zero(eng):
    """the default: the restatement's convention (depth 0 at a clipped end), not the reference's leftover"""
    import oracle.port as P
    eng.lte_track_clipper_stores(0)
    name = "shaded_flat_lines"
    steps = MG.stale_depth_cases()[name]
    c, z = _run(eng, steps, True)
    orc = P.Oracle(W, H)
    oc, oz = MG.run_wire_case(steps, orc)
    assert (z == oz).all() and (c == oc).all()
    assert S.frame_digest(c) != GOLD[name]["colour"]
