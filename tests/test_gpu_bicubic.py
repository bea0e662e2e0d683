"""Bicubic patches through the C++ mirror (M_init_bicubic / M_shade_bicubic / M_render_bicubic and
the *_object forms): normals by lt_plane_normals, shading by lt_shade_points, squares through
the GPU queue -- against vectors of the reference build and the CPU restatement."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from raytracing_amd import scenes as S

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden as MG  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "bicubic_kats.npz")
W, H = MG.WIRE_W, MG.WIRE_H


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def eng():
    import torch  # noqa: F401  (HIP runtime first, see raytracing_amd/_lib.py)
    from raytracing_amd import _lib
    _lib.lib()
    L = C.CDLL(os.path.join(os.path.dirname(_lib.SO_PATH), "liblt_engine.so"))
    L.lte_cbuf.restype = C.POINTER(C.c_int32)
    L.lte_zbuf.restype = C.POINTER(C.c_int32)
    L.lte_init(W, H)
    tex = np.ascontiguousarray(S.make_texture(9, 6), dtype=np.int32)
    L.lte_set_texture(0, _p(tex), 6)
    L._keep = tex
    return L


def mirror_patch(eng, case, as_object):
    name, c, lrs, rt, pos, ang = case
    size = (1 << lrs) + 1
    normals = np.zeros((size * size, 4), np.int32)
    inten = np.zeros((size * size, 3), np.int32)
    cc, li = np.ascontiguousarray(c, dtype=np.int32), MG.BICUBIC_LIGHTS.copy()
    eng.lte_clear(0x000020, 255, 255, 255)
    eng.lte_bicubic(_p(cc), lrs, 0xFFFFFF, 7, 0 if rt == S.M_TEXTURED else -1, *pos, *ang, _p(li), li.shape[0], rt, 7,
                    int(as_object), _p(normals), _p(inten))
    eng.lte_flush()
    c_ = np.ctypeslib.as_array(eng.lte_cbuf(), shape=(H, W)).copy()
    z_ = np.ctypeslib.as_array(eng.lte_zbuf(), shape=(H, W)).copy()
    return normals[:, :3].copy(), inten, c_, z_


@pytest.mark.parametrize("case", MG.bicubic_cases(), ids=lambda c: c[0])
def test_patch_objects_match_reference_vectors(eng, case):
    import oracle.port as P
    g = np.load(GOLD)
    n, i, c, z = mirror_patch(eng, case, as_object=True)
    assert np.array_equal(n, g[case[0] + "_normals"]) and np.array_equal(i, g[case[0] + "_intensities"])
    on, oi, oc, oz = MG.run_bicubic_case(case, P.Oracle(W, H))
    assert np.array_equal(c, oc) and np.array_equal(z, oz)
    assert [S.frame_digest(c), S.frame_digest(z)] == list(g[case[0] + "_digests"])


def test_single_patch_near_the_camera(eng):
    """M_render_bicubic on its own leaves M_force_linear_tmapping alone (0 here); squares cross the
    near plane and the screen borders"""
    import oracle.port as P
    c = S.gen_bicubic_patch(103, extent=500, bulge=300, centre=(30, 10, 330))
    for rt in (S.M_SHADED, S.M_TEXTURED, S.M_WIRE):
        case = ("x", c, 4, rt, (0, 0, 20), (9, 5, 250))
        n, i, cc, zz = mirror_patch(eng, case, as_object=False)
        o = P.Oracle(W, H)
        o.set_texture(0, S.make_texture(9, 6), 6)
        o.clear(0x000020, 255, 255, 255)
        on, oi = o.bicubic(c, 4, 0xFFFFFF, 7, 0 if rt == S.M_TEXTURED else -1, (0, 0, 20), (9, 5, 250), MG.BICUBIC_LIGHTS, rt, 7, 0)
        assert np.array_equal(n, on[:, :3]) and np.array_equal(i, oi)
        assert np.array_equal(cc, o.cbuf()) and np.array_equal(zz, o.zbuf())
        assert (cc != cc[0, 0]).sum() > 1000


def test_shade_points_entry(eng):
    """lt_shade_points on its own context, host arrays.  Checker: the restatement's polygon
    shading (M_shade_polygon over the same M_shade_vertex) run with an identity rotation on
    points and on normals rotated beforehand."""
    import oracle.port as P
    from raytracing_amd import gpu as G
    c = S.gen_bicubic_patch(7)
    o = P.Oracle(W, H)
    o.clear(0, 0, 0, 0)
    ang, pos = (40, 7, 211), (120, -60, 900)
    normals, _ = o.bicubic(c, 3, 0, 7, -1, pos, ang, MG.BICUBIC_LIGHTS, S.M_SHADED, 7)
    s9 = P.self_matrix(*ang)
    pts = np.zeros((16, 4), np.int32)
    pts[:, :3] = c[:, :3] + np.array(pos, np.int32)
    nrm = np.zeros((16, 4), np.int32)
    nrm[:, :3] = normals[:16, :3]
    rotated = np.zeros_like(nrm)
    for k in range(16):
        x, y, z = [int(v) for v in nrm[k, :3]]
        for r in range(3):
            rotated[k, r] = ((int(s9[3 * r]) * x) >> 15) + ((int(s9[3 * r + 1]) * y) >> 15) + ((int(s9[3 * r + 2]) * z) >> 15)
    pv = np.zeros((16, 6), np.int32)
    pv[:, 0] = np.arange(16)
    polys = np.array([[0, -1, 15, 0]], np.int32)
    identity = np.array([1 << 15, 0, 0, 0, 1 << 15, 0, 0, 0, 1 << 15], np.int32)
    want_pv, _, skipped, bad = P.shade_polys(identity, (0, 0, 0), pts, rotated, polys, pv, np.arange(16, dtype=np.int32), None,
                                             MG.BICUBIC_LIGHTS)
    assert skipped == 0 and bad == 0
    got = G.Context(64, 64).shade_points(s9, pts, nrm, MG.BICUBIC_LIGHTS)
    assert np.array_equal(got, want_pv[:, 1:4])
