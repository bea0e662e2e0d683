"""Differential test of the CPU restatement against the reference's own code, where a
checker library built from the reference is present (oracle/_ref, see
oracle/build_ref.sh) or can be built (/root/reference present).  CPU only."""
import numpy as np
import pytest

import oracle.port as P
import oracle.ref as R
from raytracing_amd import scenes as S

W, H = 160, 120
pytestmark = pytest.mark.skipif(not R.available(W, H), reason="no reference checker library and no reference tree")


@pytest.mark.parametrize("rt", [S.M_FLAT, S.M_SHADED, S.M_TEXTURED])
@pytest.mark.parametrize("seed", range(10))
def test_fuzz_frames(rt, seed):
    ref, orc = R.Ref(W, H), P.Oracle(W, H)
    sc = S.gen_fuzz_polys(60, W, H, 7, 31 * seed + rt, render_type=rt, textured=rt == S.M_TEXTURED)
    for h, (tex, lg) in enumerate(sc.textures):
        ref.set_texture(h, tex, lg)
        orc.set_texture(h, tex, lg)
    ref.clear(0x445566, 200, 220, 240)
    orc.clear(0x445566, 200, 220, 240)
    vv = P.transform(sc.matrix, sc.verts)
    assert np.array_equal(vv[:, :3], ref.transform(sc.matrix, sc.verts)[:, :3])
    ref.render_polys(vv, sc.polys, sc.polyaux, sc.polyverts, rt, sc.log_focus)
    orc.render_polys(vv, sc.polys, sc.polyaux, sc.polyverts, rt, sc.log_focus)
    assert np.array_equal(ref.cbuf(), orc.cbuf()) and np.array_equal(ref.zbuf(), orc.zbuf())


@pytest.mark.parametrize("dim,kind", [(3, 2), (6, 4), (8, 8)])
def test_screen_polygons(dim, kind):
    ref, orc = R.Ref(W, H), P.Oracle(W, H)
    tex = S.make_texture(99, 5)
    ref.set_texture(0, tex, 5)
    orc.set_texture(0, tex, 5)
    ref.clear(0, 0, 0, 0)
    orc.clear(0, 0, 0, 0)
    for (e, t, col) in S.gen_screen_polys(150, W, H, seed=dim, dim=dim):
        ref.g_polygon(kind, t, colour=col, r=100, g=150, b=200, tex=0)
        orc.g_polygon(kind, t, colour=col, r=100, g=150, b=200, tex=0)
    assert np.array_equal(ref.cbuf(), orc.cbuf()) and np.array_equal(ref.zbuf(), orc.zbuf())


def test_rotating_object_matrix_and_frames():
    ref, orc = R.Ref(W, H), P.Oracle(W, H)
    sc = S.gen_c2(800, W, H, 7, radius=12, seed=11)
    for f in range(0, 256, 37):
        cam = (3, 250, 7, 5, -3, 10)
        obj = (f & 255, (2 * f) & 255, (3 * f) & 255, 0, 0, 600)
        m = P.make_matrix(cam, obj, 0)
        assert np.array_equal(m, ref.make_matrix(cam, obj, 7))
        vv = P.transform(m, sc.verts)
        ref.clear(0, 0, 0, 0)
        orc.clear(0, 0, 0, 0)
        ref.render_polys(vv, sc.polys, None, sc.polyverts, S.M_SHADED, 7)
        orc.render_polys(vv, sc.polys, None, sc.polyverts, S.M_SHADED, 7)
        assert np.array_equal(ref.cbuf(), orc.cbuf()) and np.array_equal(ref.zbuf(), orc.zbuf())
