"""Edge cases of the HIP path through the C-ABI: empty and tiny inputs, the largest
polygons, limits that must be reported and not crash, extreme screen shapes."""
import numpy as np
import pytest

from raytracing_amd import scenes as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import oracle.port as P
    from raytracing_amd import gpu as G
    return P, G


def _ngon(n, cx, cy, r, z, lf, w, h, rgb=(200, 100, 50)):
    """one convex n-gon facing the viewer, in view space -> (verts, polys, polyverts)"""
    ang = -np.arange(n) * 2 * np.pi / n               # clockwise on screen = projected cross product < 0
    px = np.round(cx + r * np.cos(ang)).astype(np.int64)
    py = np.round(cy + r * np.sin(ang)).astype(np.int64)
    verts = np.zeros((n, 4), np.int32)
    verts[:, 0] = ((px - w // 2) * z) >> lf
    verts[:, 1] = ((py - h // 2) * z) >> lf
    verts[:, 2] = z
    pv = np.zeros((n + 1, 6), np.int32)
    pv[:n, 0] = np.arange(n)
    pv[:n, 1:4] = np.array(rgb)[None, :] + (np.arange(n)[:, None] * 3)
    pv[n] = pv[0]
    polys = np.array([[0xFFFFFF, -1, n, 0]], np.int32)
    return verts, polys, pv


def _both(P, G, w, h, lf, verts, polys, pv, rt=S.M_SHADED, aux=None, max_edges=0):
    ctx = G.Context(w, h)
    orc = P.Oracle(w, h)
    orc.clear(0x334455, 255, 255, 255)
    orc.render_polys(verts, polys, aux, pv, rt, lf)
    ctx.clear(P.cl_light(0x334455, 255, 255, 255))
    ctx.submit_object(verts, polys, pv, rt, lf, polyaux=aux, max_edges=max_edges)
    return ctx, orc


def test_empty_submit_and_clear_only(env):
    P, G = env
    ctx = G.Context(100, 60)
    ctx.clear(0x123456)
    ctx.submit_object(np.zeros((1, 4), np.int32), np.zeros((0, 4), np.int32), np.zeros((1, 6), np.int32), S.M_SHADED, 8)
    c, z = ctx.readback()
    assert (c == 0x123456).all() and (z == 2**31 - 1).all()


def test_draw_before_first_clear_is_an_error(env):
    P, G = env
    ctx = G.Context(64, 64)
    v, po, pv = _ngon(3, 30, 30, 10, 500, 7, 64, 64)
    with pytest.raises(G.LtError):
        ctx.submit_object(v, po, pv, S.M_SHADED, 7)
    with pytest.raises(G.LtError):
        ctx.readback()


@pytest.mark.parametrize("n", [3, 4, 7, 15])
def test_single_polygons_up_to_fifteen_edges(env, n):
    P, G = env
    v, po, pv = _ngon(n, 90, 70, 55, 400, 7, 200, 150)
    ctx, orc = _both(P, G, 200, 150, 7, v, po, pv, max_edges=15)
    c, z = ctx.readback()
    assert (z == orc.zbuf()).all() and (c == orc.cbuf()).all()
    assert ctx.counters()["fragments"] == orc.counters()["fragments"] > 0


def test_fifteen_gon_clipped_by_two_borders(env):
    """clipping adds tuples: up to 15 + 2 per border; the reference's arrays hold 16"""
    P, G = env
    v, po, pv = _ngon(12, 10, 8, 60, 400, 7, 200, 150)
    ctx, orc = _both(P, G, 200, 150, 7, v, po, pv, max_edges=15)
    c, z = ctx.readback()
    assert (z == orc.zbuf()).all() and (c == orc.cbuf()).all()


def test_more_edges_than_declared_is_reported_not_drawn(env):
    P, G = env
    v, po, pv = _ngon(9, 90, 70, 55, 400, 7, 200, 150)
    ctx = G.Context(200, 150)
    ctx.clear(0)
    ctx.submit_object(v, po, pv, S.M_SHADED, 7, max_edges=5)
    with pytest.raises(G.LtError) as ei:
        ctx.flush()
    assert ei.value.code == -3
    c, z = ctx.readback(check=False)
    assert (z == 2**31 - 1).all()
    assert ctx.counters()["overflow_polys"] == 1


def test_bad_indices_are_reported_not_read(env):
    P, G = env
    v, po, pv = _ngon(3, 90, 70, 30, 400, 7, 200, 150)
    pv = pv.copy()
    pv[1, 0] = 1 << 30                                 # vertex index far out of range
    po = np.concatenate([po, np.array([[0xFFFFFF, -1, 3, 1 << 29]], np.int32)])      # records out of range
    ctx = G.Context(200, 150)
    ctx.clear(0)
    ctx.submit_object(v, po, pv, S.M_SHADED, 7, max_edges=3)
    with pytest.raises(G.LtError) as ei:
        ctx.flush()
    assert ei.value.code == -3 and ctx.counters()["overflow_polys"] == 2


def test_tile_list_capacity_overflow_is_reported_and_recoverable(env):
    P, G = env
    big = S.gen_c2(80000, 640, 480, 9, radius=600, seed=12)       # every triangle spans most of the 150 tiles
    ctx = G.Context(big.width, big.height)
    ctx.set_list_capacity(1 << 20)
    ctx.clear(0)
    ctx.submit_scene(big)
    with pytest.raises(G.LtError) as ei:
        ctx.flush()
    assert ei.value.code == -4
    assert ctx.counters()["list_overflow"] > 0
    ctx.set_list_capacity(0)                           # back to the default sizing
    sc = S.gen_c2(3000, 640, 480, 9, radius=30, seed=13)
    ctx.clear(0)
    ctx.reset_counters()
    ctx.submit_scene(sc)
    c, z = ctx.readback()
    orc = P.Oracle(sc.width, sc.height)
    orc.clear(0, 0, 0, 0)
    orc.render_polys(P.transform(sc.matrix, sc.verts), sc.polys, None, sc.polyverts, sc.render_type, sc.log_focus)
    assert (z == orc.zbuf()).all() and (c == orc.cbuf()).all()


@pytest.mark.parametrize("w,h", [(2, 2), (3, 200), (4096, 5), (16384, 33), (70, 16384), (16384, 1100)])   # the last: beyond the LDS histogram of the binning
def test_extreme_screen_shapes(env, w, h):
    P, G = env
    sc = S.gen_c2(400, w, h, 8, radius=max(2, min(w, h) // 3), seed=w * 7 + h)
    ctx = G.Context(w, h)
    orc = P.Oracle(w, h)
    orc.clear(0x0a0b0c, 255, 255, 255)
    orc.render_polys(P.transform(sc.matrix, sc.verts), sc.polys, None, sc.polyverts, sc.render_type, sc.log_focus)
    ctx.clear(P.cl_light(0x0a0b0c, 255, 255, 255))
    ctx.submit_scene(sc)
    c, z = ctx.readback()
    assert (z == orc.zbuf()).all() and (c == orc.cbuf()).all()
    assert ctx.counters()["fragments"] == orc.counters()["fragments"]


def test_degenerate_polygons_draw_nothing_or_as_the_reference(env):
    P, G = env
    lf, w, h = 7, 200, 150
    verts = np.array([[10, 10, 400, 0], [10, 10, 400, 0], [10, 10, 400, 0],          # a point
                      [-50, 0, 400, 0], [60, 0, 400, 0], [5, 0, 400, 0],             # a horizontal line
                      [0, -40, 400, 0], [0, 50, 400, 0], [0, 5, 400, 0]], np.int32)  # a vertical line
    pv = np.zeros((12, 6), np.int32)
    for p in range(3):
        pv[4 * p:4 * p + 3, 0] = np.arange(3) + 3 * p
        pv[4 * p + 3, 0] = 3 * p
    pv[:, 1:4] = 128
    polys = np.array([[0xFFFFFF, -1, 3, 0], [0xFFFFFF, -1, 3, 4], [0xFFFFFF, -1, 3, 8]], np.int32)
    ctx, orc = _both(P, G, w, h, lf, verts, polys, pv)
    c, z = ctx.readback()
    assert (z == orc.zbuf()).all() and (c == orc.cbuf()).all()
