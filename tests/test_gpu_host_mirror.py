"""The C++ host mirror of the reference's API (raytracing_amd/host/lt_engine.h) driving
the HIP library: M_render_polygon_object, M_render_polygon and G_*_polygon called as a
reference-style program would, checked against the CPU checker."""
import ctypes as C
import os

import numpy as np
import pytest

from raytracing_amd import scenes as S

pytestmark = pytest.mark.gpu
W, H = 200, 150


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def eng():
    import torch  # noqa: F401  (HIP runtime first, see raytracing_amd/_lib.py)
    from raytracing_amd import _lib
    _lib.lib()
    so = os.path.join(os.path.dirname(_lib.SO_PATH), "liblt_engine.so")
    L = C.CDLL(so)
    L.lte_cbuf.restype = C.POINTER(C.c_int32)
    L.lte_zbuf.restype = C.POINTER(C.c_int32)
    L.lte_init(W, H)
    return L


def _frame(L):
    L.lte_flush()
    c = np.ctypeslib.as_array(L.lte_cbuf(), shape=(H, W)).copy()
    z = np.ctypeslib.as_array(L.lte_zbuf(), shape=(H, W)).copy()
    return c, z


@pytest.mark.parametrize("rt", [S.M_SHADED, S.M_FLAT, S.M_TEXTURED])
def test_render_polygon_object(eng, rt):
    import oracle.port as P
    sc = S.gen_fuzz_polys(120, W, H, 7, 4000 + rt, render_type=rt, textured=rt == S.M_TEXTURED, equal_lights=False)
    assert sc.verts.shape[0] < 1024
    cam = (5, 250, 3, 10, -5, -40)
    obj = (7, 9, 250, 3, 4, 60)
    tex = None
    orc = P.Oracle(W, H)
    if sc.textures:
        tex = np.ascontiguousarray(sc.textures[0][0], dtype=np.int32)
        eng.lte_set_texture(0, _p(tex), sc.textures[0][1])
        orc.set_texture(0, tex, sc.textures[0][1])
    orc.clear(0x203040, 250, 240, 230)
    m = P.make_matrix(cam, obj, 0)
    orc.render_polys(P.transform(m, sc.verts), sc.polys, sc.polyaux, sc.polyverts, rt, sc.log_focus)
    eng.lte_clear(0x203040, 250, 240, 230)
    v, po, pa, pv = sc.verts.copy(), sc.polys.copy(), sc.polyaux.copy(), sc.polyverts.copy()
    eng.lte_render_object(_p(v), v.shape[0], _p(po), _p(pa), _p(pv), po.shape[0], rt, *cam, sc.log_focus, *obj[3:], *obj[:3])
    c, z = _frame(eng)
    assert np.array_equal(z, orc.zbuf()) and np.array_equal(c, orc.cbuf())


def test_render_polygon_one_by_one_and_screen_polys(eng):
    import oracle.port as P
    sc = S.gen_c2(1500, W, H, 7, radius=14, seed=21)
    orc = P.Oracle(W, H)
    orc.clear(0, 0, 0, 0)
    obj = (3, 5, 7, 2, -3, 20)
    m = P.make_matrix((0, 0, 0, 0, 0, 0), obj, 0)
    orc.render_polys(P.transform(m, sc.verts), sc.polys, None, sc.polyverts, S.M_SHADED, sc.log_focus)
    polys = S.gen_screen_polys(60, W, H, seed=5, dim=6)
    for (e, t, col) in polys:
        orc.g_polygon(4, t, colour=col)
    eng.lte_clear(0, 0, 0, 0)
    v, po, pv = sc.verts.copy(), sc.polys.copy(), sc.polyverts.copy()
    eng.lte_render_polys_one_by_one(_p(v), v.shape[0], _p(po), None, _p(pv), po.shape[0], S.M_SHADED, sc.log_focus,
                                    *obj[3:], *obj[:3])
    for (e, t, col) in polys:
        tt = np.ascontiguousarray(t, dtype=np.int32)
        eng.lte_g_polygon(4, _p(tt), e, col, 0, 0, 0, 0)
    c, z = _frame(eng)
    assert np.array_equal(z, orc.zbuf()) and np.array_equal(c, orc.cbuf())


def test_shade_polygon_object(eng):
    """M_shade_polygon_object through the mirror: records and flat colours written back
    into the caller's polygons, equal to the restatement's"""
    import oracle.port as P
    case = S.gen_shading_case(200, 77, n_verts=600, n_normals=300)
    ang, pos = (40, 7, 211), (120, -60, 900)
    want_pv, want_pa, skipped, bad = P.shade_polys(P.self_matrix(*ang), pos, case["verts"], case["normals"],
                                                   case["polys"], case["polyverts"], case["polynormals"], None,
                                                   case["lights"])
    assert skipped == 0 and bad == 0
    v, n, po = case["verts"].copy(), case["normals"].copy(), case["polys"].copy()
    pv, pn = case["polyverts"].copy(), case["polynormals"].copy()
    aux = np.zeros((po.shape[0], 4), np.int32)
    li = case["lights"].copy()
    eng.lte_shade_object(_p(v), v.shape[0], _p(n), n.shape[0], _p(po), _p(aux), _p(pv), _p(pn), po.shape[0],
                         *pos, *ang, _p(li), li.shape[0])
    assert np.array_equal(pv, want_pv) and np.array_equal(aux[:, :3], want_pa[:, :3])


def test_texels_changed_in_place_are_seen(eng):
    """The reference reads texel memory at draw time: the same array with new contents in the
    next frame draws the new texels (the mirror knows an uploaded texture by its address)."""
    import oracle.port as P
    sc = S.gen_fuzz_polys(60, W, H, 7, 4321, render_type=S.M_TEXTURED, textured=True, equal_lights=True)
    tex = np.ascontiguousarray(sc.textures[0][0], dtype=np.int32).copy()
    lg = sc.textures[0][1]
    eng.lte_set_texture(1, _p(tex), lg)
    sc.polys[:, 1] = 1
    cam, obj = (0, 0, 0, 0, 0, 0), (0, 0, 0, 0, 0, 30)
    m = P.make_matrix(cam, obj, 0)
    for rnd in range(2):
        orc = P.Oracle(W, H)
        orc.set_texture(1, tex, lg)
        orc.clear(0, 0, 0, 0)
        orc.render_polys(P.transform(m, sc.verts), sc.polys, sc.polyaux, sc.polyverts, S.M_TEXTURED, sc.log_focus)
        eng.lte_clear(0, 0, 0, 0)
        v, po, pa, pv = sc.verts.copy(), sc.polys.copy(), sc.polyaux.copy(), sc.polyverts.copy()
        eng.lte_render_object(_p(v), v.shape[0], _p(po), _p(pa), _p(pv), po.shape[0], S.M_TEXTURED, *cam, sc.log_focus,
                              *obj[3:], *obj[:3])
        c, z = _frame(eng)
        assert np.array_equal(z, orc.zbuf()) and np.array_equal(c, orc.cbuf()), rnd
        tex[:] = S.make_texture(99, lg)             # same address, other texels


def test_more_texture_arrays_than_handles(eng):
    """The reference knows a texture by its pointer and has no limit on their number; the library has
    LT_MAX_TEXTURES handles.  The mirror hands the handle of the address unused for longest to a new
    address (submitting the open batch first when every handle belongs to it): 150 arrays used by screen-level
    polygons in one frame, then objects whose polygons use four of the earlier arrays each."""
    import oracle.port as P
    lg = 4
    texs = [np.ascontiguousarray(S.make_texture(1000 + i, lg), dtype=np.int32) for i in range(150)]
    polys = S.gen_screen_polys(150, W, H, seed=61, dim=8)
    orc = P.Oracle(W, H)
    orc.clear(0x101010, 255, 255, 255)
    eng.lte_clear(0x101010, 255, 255, 255)
    keep = []
    for i, (e, t, col) in enumerate(polys):
        slot = i % 64
        orc.set_texture(slot, texs[i], lg)
        orc.g_polygon(8, t, colour=col, tex=slot)
        tt = np.ascontiguousarray(t, dtype=np.int32)
        keep.append(tt)
        eng.lte_set_texture(slot, _p(texs[i]), lg)
        eng.lte_g_polygon(8, _p(tt), e, col, 0, 0, 0, slot)
    for k in range(3):
        sc = S.gen_fuzz_polys(80, W, H, 7, 4400 + k, render_type=S.M_TEXTURED, textured=True, equal_lights=True)
        sc.polys[:, 1] = np.arange(sc.polys.shape[0]) % 4
        obj = (k, 3, 5, 0, 0, 25 + 5 * k)
        for j in range(4):
            orc.set_texture(j, texs[10 * k + j], lg)
            eng.lte_set_texture(j, _p(texs[10 * k + j]), lg)
        m = P.make_matrix((0, 0, 0, 0, 0, 0), obj, 0)
        orc.render_polys(P.transform(m, sc.verts), sc.polys, sc.polyaux, sc.polyverts, S.M_TEXTURED, sc.log_focus)
        v, po, pa, pv = sc.verts.copy(), sc.polys.copy(), sc.polyaux.copy(), sc.polyverts.copy()
        keep += [v, po, pa, pv]
        eng.lte_render_object(_p(v), v.shape[0], _p(po), _p(pa), _p(pv), po.shape[0], S.M_TEXTURED, 0, 0, 0, 0, 0, 0, sc.log_focus,
                              *obj[3:], *obj[:3])
    c, z = _frame(eng)
    assert np.array_equal(z, orc.zbuf()) and np.array_equal(c, orc.cbuf())


def test_many_objects_become_tables_and_the_frame_is_that_of_the_calls_in_turn(eng):
    """M_render_polygon_object forty times over one frame -- runs of Gouraud, flat, textured and Gouraud objects, every
    object with its own place and angles: the mirror turns each run into one table of objects (one GPU batch).
    Against the reference build's own M_render_polygon_object, call by call, where oracle/_ref has it for this screen."""
    import oracle.port as P
    import oracle.ref as R
    use_ref = os.path.exists(R.so_path(W, H))
    chk = R.Ref(W, H) if use_ref else P.Oracle(W, H)
    cam = (3, 251, 2, 12, -6, -30)
    lf = 7
    tex = np.ascontiguousarray(S.make_texture(77, 6), dtype=np.int32)
    eng.lte_set_texture(0, _p(tex), 6)
    chk.set_texture(0, tex, 6)
    chk.clear(0x304050, 255, 250, 245)
    eng.lte_clear(0x304050, 255, 250, 245)
    if use_ref:
        chk.make_matrix(cam, (0, 0, 0, 0, 0, 0), lf)       # M_set_camera
    keep = []
    runs = [(S.M_SHADED, 14), (S.M_FLAT, 9), (S.M_TEXTURED, 10), (S.M_SHADED, 7)]
    k = 0
    for (rt, n) in runs:
        for _ in range(n):
            if rt == S.M_TEXTURED:
                # (regular texture mappings: on a singular one the reference divides by zero, TransBs.cpp:505)
                sc = S.gen_c3(60 + k % 40, W, H, lf, radius=14, seed=5000 + k, equal_lights=True, lin_percent=70, regular_mapping=True)
            else:
                sc = S.gen_fuzz_polys(70 + k % 40, W, H, lf, 5000 + k, render_type=rt, max_edges=3 + k % 4)
            assert sc.verts.shape[0] < 1024 and sc.polys.shape[0] <= 256
            pos = (k % 5 - 2, 3 - k % 7, 20 + 9 * (k % 11))
            ang = ((5 * k) % 256 if k % 4 == 0 else 0, (250 + k) % 256 if k % 3 == 0 else 0, k % 6)
            v, po, pa, pv = sc.verts.copy(), sc.polys.copy(), sc.polyaux.copy(), sc.polyverts.copy()
            keep.append((v, po, pa, pv))
            if use_ref:
                chk.render_object(v, po, pa, pv, rt, pos, ang)
            else:
                m = P.make_matrix(cam, ang + pos, 0)
                chk.render_polys(P.transform(m, v), po, pa, pv, rt, lf)
            eng.lte_render_object(_p(v), v.shape[0], _p(po), _p(pa), _p(pv), po.shape[0], rt, *cam, lf, *pos, *ang)
            k += 1
    c, z = _frame(eng)
    assert np.array_equal(z, chk.zbuf()), f"depth differs at {int((z != chk.zbuf()).sum())} pixels"
    assert np.array_equal(c, chk.cbuf()), f"colour differs at {int((c != chk.cbuf()).sum())} pixels"
