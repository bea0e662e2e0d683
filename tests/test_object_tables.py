"""Host side of the tables of objects (raytracing_amd.objects), no GPU: a triangle soup cut into objects and
joined again is the soup; the CPU restatement drawing the objects one call at a time -- what a program written
against M_render_polygon_object (EngPoly.cpp:499-525) does -- leaves the frame of the soup drawn whole."""
import numpy as np
import pytest

from raytracing_amd import objects as O, scenes as S


def test_split_and_join_give_the_soup_back():
    sc = S.gen_c2(1000, 320, 200, 8, radius=12, seed=31)
    parts = O.split_soup(sc.verts, sc.polys, sc.polyverts, 64)
    assert len(parts) == (1000 + 63) // 64
    for v, p, pv, aux in parts:
        assert aux is None
        assert p[0, 3] == 0 and pv[:, 0].min() == 0 and pv[:, 0].max() == v.shape[0] - 1
        assert p.shape[0] <= 64 and v.shape[0] < 1024                    # the reference's limits for one call (Engine.h:432-433)
    tab = O.build_table([(sc.matrix, v, p, pv, aux) for v, p, pv, aux in parts])
    assert tab.objects.shape == (len(parts), O.OBJECT_INTS) and tab.objects.dtype == np.int32
    assert np.array_equal(tab.verts, sc.verts) and np.array_equal(tab.polyverts[:, 1:], sc.polyverts[:, 1:])
    assert np.array_equal(tab.polys, sc.polys)
    assert np.array_equal(tab.polyobj, np.arange(1000) // 64)
    # vertex ranges follow each other; the records' indices count from their object's first vertex
    assert tab.objects[0, 12] == 0
    assert np.array_equal(tab.objects[1:, 12], np.cumsum(tab.objects[:-1, 13]))
    assert np.array_equal(tab.polyverts[:, 0] + tab.objects[tab.polyobj[np.searchsorted(tab.polys[:, 3], np.arange(tab.polyverts.shape[0]), side="right") - 1], 12],
                          sc.polyverts[:, 0])
    assert all(np.array_equal(tab.objects[i, :12], np.asarray(sc.matrix).reshape(12)) for i in range(len(parts)))
    assert tab.polyaux is None and tab.max_edges == 3


def test_objects_drawn_in_turn_leave_the_frame_of_the_soup():
    import oracle.port as P
    sc = S.gen_fuzz_polys(300, 160, 120, 7, 41, render_type=S.M_SHADED)
    whole = P.Oracle(sc.width, sc.height)
    whole.clear(0x102030, 255, 255, 255)
    whole.render_polys(P.transform(sc.matrix, sc.verts), sc.polys, sc.polyaux, sc.polyverts, sc.render_type, sc.log_focus, sc.force_linear)
    # the same polygons as objects of 25 polygons over the shared vertex array, each with a matrix of its own that
    # undoes a shift of its vertices (so that the matrices matter)
    by_turn = P.Oracle(sc.width, sc.height)
    by_turn.clear(0x102030, 255, 255, 255)
    objs = []
    for k, a in enumerate(range(0, sc.n_polys, 25)):
        b = min(a + 25, sc.n_polys)
        shift = np.array([7 * k, -3 * k, 5 * k, 0], np.int32)
        m = np.asarray(sc.matrix, np.int32).reshape(12).copy()
        m[3] -= shift[0]; m[7] -= shift[1]; m[11] -= shift[2]
        polys = sc.polys[a:b].copy()
        f0 = int(polys[0, 3])
        f1 = int(polys[-1, 3] + polys[-1, 2] + 1)
        polys[:, 3] -= f0
        objs.append((m, sc.verts + shift, polys, sc.polyverts[f0:f1].copy(), None if sc.polyaux is None else sc.polyaux[a:b].copy()))
    if not np.array_equal(np.asarray(sc.matrix).reshape(12)[[0, 1, 2, 4, 5, 6, 8, 9, 10]], np.array([1 << 15, 0, 0, 0, 1 << 15, 0, 0, 0, 1 << 15])):
        pytest.skip("the scene's matrix is not a translation: the shift does not commute with it")
    for m, v, p, pv, aux in objs:
        by_turn.render_polys(P.transform(m, v), p, aux, pv, sc.render_type, sc.log_focus, sc.force_linear)
    assert np.array_equal(by_turn.cbuf(), whole.cbuf()) and np.array_equal(by_turn.zbuf(), whole.zbuf())
    assert by_turn.counters()["fragments"] == whole.counters()["fragments"]
    tab = O.build_table(objs)
    assert tab.polys.shape[0] == sc.n_polys and tab.objects.shape[0] == len(objs)
    assert np.array_equal(tab.objects[:, 12], np.arange(len(objs)) * sc.verts.shape[0])
