"""Object set-up on the GPU: the plane normals of M_init_polygon_object (lt_plane_normals[_dev])
and the host mirror's M_init_polygon_object, against the CPU restatement and the vectors the
reference build produced; then set up, shaded and drawn in one go."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from raytracing_amd import scenes as S

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden as MG  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "normals_kats.npz")


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def env():
    import oracle.port as P
    from raytracing_amd import gpu as G
    return P, G


def soup_planes(P, sc):
    """T_normal_plane of every triangle of a soup by the restatement; None where it stops"""
    out = np.zeros((sc.n_polys, 4), np.int32)
    zero = 0
    for i in range(sc.n_polys):
        f = sc.polys[i, 3]
        idx = sc.polyverts[f:f + 3, 0]
        n = P.normal_plane(sc.verts[idx[0], :3], sc.verts[idx[1], :3], sc.verts[idx[2], :3])
        if n is None:
            zero += 1
        else:
            out[i, :3] = n
    return out, zero


@pytest.mark.parametrize("scale", [1, 300])
def test_plane_normals_of_a_soup(env, scale):
    """device arrays; scale 300 puts coordinates beyond 16 bits: the vector product wraps and
    T_norm takes its long-vector branch"""
    import torch
    P, G = env
    sc = S.gen_c2(6000, 640, 480, 9, radius=8, seed=71)
    sc.verts[:, :3] *= scale
    want, zero = soup_planes(P, sc)
    ctx = G.Context(64, 64)
    d_v, d_p, d_pv = (torch.from_numpy(a).cuda() for a in (sc.verts, sc.polys, sc.polyverts))
    d_out = torch.full((sc.n_polys, 4), -7, dtype=torch.int32, device="cuda")
    ctx.plane_normals_dev(d_v, d_p, d_pv, d_out)
    if zero:
        with pytest.raises(G.LtError):
            ctx.flush()
    else:
        ctx.flush()
    assert np.array_equal(d_out.cpu().numpy(), want)


def test_zero_length_normal_is_reported(env):
    P, G = env
    ob = S.gen_sphere_object(6, 4, 100)
    bad = ob["verts"].copy()
    bad[ob["polyverts"][1, 0], :3] = bad[ob["polyverts"][0, 0], :3]
    ctx = G.Context(64, 64)
    with pytest.raises(G.LtError, match="zero"):
        ctx.plane_normals(bad, ob["polys"], ob["polyverts"])
    out = ctx.plane_normals(ob["verts"], ob["polys"], ob["polyverts"])          # the context goes on working
    assert (np.abs(out[:, :3]).sum(axis=1) > 0).all() and (out[:, 3] == 0).all()


@pytest.fixture(scope="module")
def eng():
    import torch  # noqa: F401  (HIP runtime first, see raytracing_amd/_lib.py)
    from raytracing_amd import _lib
    _lib.lib()
    L = C.CDLL(os.path.join(os.path.dirname(_lib.SO_PATH), "liblt_engine.so"))
    L.lte_cbuf.restype = C.POINTER(C.c_int32)
    L.lte_zbuf.restype = C.POINTER(C.c_int32)
    L.lte_init_object.restype = C.c_int
    L.lte_init(200, 150)
    return L


def mirror_init(eng, ob, ty):
    v, po, pv = ob["verts"].copy(), ob["polys"].copy(), ob["polyverts"].copy()
    normals = np.zeros((1024, 4), np.int32)
    pn = np.full(pv.shape[0], -2, np.int32)
    nn = eng.lte_init_object(_p(v), v.shape[0], _p(po), _p(np.ascontiguousarray(ty, dtype=np.int32)), _p(pv), po.shape[0],
                             _p(normals), _p(pn))
    return normals[:nn], pn


def test_init_polygon_object_matches_reference_vectors(eng):
    g = np.load(GOLD)
    for k, c in enumerate(MG.NORMAL_CASES):
        ob = S.gen_sphere_object(c["n_long"], c["n_lat"], c["radius"], c["centre"])
        ty = MG.normal_case_types(ob["types"], c["mix"])
        n, pn = mirror_init(eng, ob, ty)
        assert np.array_equal(n[:, :3], g[f"normals_{k}"])
        assert np.array_equal(pn, g[f"polynormals_{k}"])


def test_init_shade_render_chain(eng):
    """M_init_polygon_object -> M_shade_polygon_object -> M_render_polygon_object through the
    mirror, against the same chain of the CPU restatement"""
    import oracle.port as P
    W, H = 200, 150
    ob = S.gen_sphere_object(16, 8, 120, (0, 0, 0))
    ty = ob["types"]
    want_n, want_pn = P.init_normals(ob["verts"], ob["polys"], ty, ob["polyverts"])
    n, pn = mirror_init(eng, ob, ty)
    assert np.array_equal(n[:, :3], want_n) and np.array_equal(pn, want_pn)
    lights = np.array([[1, 0, 0, 0, 0, 30, 30, 30], [2, 400, -500, -300, 0, 220, 200, 180]], np.int32)
    ang, pos = (20, 40, 0), (10, -20, 420)
    n4 = np.zeros((want_n.shape[0], 4), np.int32)
    n4[:, :3] = want_n
    want_pv, want_pa, skipped, bad = P.shade_polys(P.self_matrix(*ang), pos, ob["verts"], n4, ob["polys"], ob["polyverts"],
                                                   want_pn, None, lights)
    assert skipped == 0 and bad == 0
    orc = P.Oracle(W, H)
    orc.clear(0, 0, 0, 0)
    m = P.make_matrix((0, 0, 0, 0, 0, 0), (*ang, *pos), 0)
    orc.render_polys(P.transform(m, ob["verts"]), ob["polys"], want_pa, want_pv, S.M_SHADED, 8)

    v, po, pv = ob["verts"].copy(), ob["polys"].copy(), ob["polyverts"].copy()
    aux = np.zeros((po.shape[0], 4), np.int32)
    nn, pnn, li = n.copy(), pn.copy(), lights.copy()
    eng.lte_shade_object(_p(v), v.shape[0], _p(nn), nn.shape[0], _p(po), _p(aux), _p(pv), _p(pnn), po.shape[0],
                         *pos, *ang, _p(li), li.shape[0])
    assert np.array_equal(pv, want_pv)
    eng.lte_clear(0, 0, 0, 0)
    eng.lte_render_object(_p(v), v.shape[0], _p(po), _p(aux), _p(pv), po.shape[0], S.M_SHADED, 0, 0, 0, 0, 0, 0, 8,
                          *pos, *ang)
    eng.lte_flush()
    c = np.ctypeslib.as_array(eng.lte_cbuf(), shape=(H, W)).copy()
    z = np.ctypeslib.as_array(eng.lte_zbuf(), shape=(H, W)).copy()
    assert (z != 2 ** 31 - 1).sum() > 2000                                      # the sphere is on screen
    assert np.array_equal(z, orc.zbuf()) and np.array_equal(c, orc.cbuf())
