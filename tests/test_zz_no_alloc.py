"""lt_submit_object_dev_noalloc (the entry point that allocates nothing: its work arrays are
pieces of one static device array) against lt_submit_object_dev on the same inputs, through
the C-ABI: frames and counters bit for bit, on two calls in a row, the second one smaller."""
import numpy as np
import pytest

from raytracing_amd import scenes as S

pytestmark = pytest.mark.gpu


def _dev(sc):
    import torch
    d = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (sc.verts, sc.polys, sc.polyverts)]
    aux = None if sc.polyaux is None else torch.from_numpy(np.ascontiguousarray(sc.polyaux)).cuda()
    return d, aux


def _draw(ctx, sc, noalloc):
    (v, po, pv), aux = _dev(sc)
    for h, (tex, lg) in enumerate(sc.textures):
        ctx.texture_upload(h, tex, lg)
    me = int(sc.polys[:, 2].max())
    f = ctx.submit_object_dev_noalloc if noalloc else ctx.submit_object_dev
    f(v, po, pv, sc.render_type, sc.log_focus, d_polyaux=aux, matrix=sc.matrix, force_linear=sc.force_linear, max_edges=me)
    try:
        ctx.flush()
    except ctx_error() as e:                # a singular texture mapping is reported, the same way by both
        return e.code
    return 0


def ctx_error():
    from raytracing_amd import gpu as G
    return G.LtError


def _frames(scenes, noalloc, w, h, overlap=True):
    """one frame per scene, each cleared first; -> [(colour, depth, counters)]"""
    from raytracing_amd import gpu as G
    ctx = G.Context(w, h)
    ctx.set_frame_overlap(overlap)
    out = []
    for k, sc in enumerate(scenes):
        ctx.clear(0x102030 + k)
        code = _draw(ctx, sc, noalloc)
        c, z = ctx.readback()
        out.append((c, z, dict(ctx.counters(), flush_code=code)))
    ctx.close()
    return out


def _same(a, b):
    for (c0, z0, n0), (c1, z1, n1) in zip(a, b):
        assert np.array_equal(z0, z1) and np.array_equal(c0, c1)
        assert n0 == n1


def test_two_calls_second_smaller_shaded():
    big = S.gen_c2(40000, 640, 480, 9, radius=10, seed=11)
    small = S.gen_c2(700, 640, 480, 9, radius=40, seed=12)
    _same(_frames([big, small], True, 640, 480), _frames([big, small], False, 640, 480))


def test_few_large_polygons_fill_the_tile_lists_alike():
    """a few thousand polygons that each cover many tiles (about 100 000 tile references: more than the
    smaller list minimum held); the flush code is part of the comparison, and so is the CPU checker's frame"""
    import oracle.port as P
    big = S.gen_c2(14000, 1920, 1080, 10, radius=220, seed=31)
    small = S.gen_c2(300, 1920, 1080, 10, radius=200, seed=32)
    a, b = _frames([big, small], True, 1920, 1080), _frames([big, small], False, 1920, 1080)
    _same(a, b)
    assert a[0][2]["flush_code"] == 0 and a[0][2]["tile_refs"] > (1 << 16)
    orc = P.Oracle(1920, 1080)
    orc.clear(0x102030, 255, 255, 255)
    orc.render_polys(P.transform(big.matrix, big.verts), big.polys, None, big.polyverts, big.render_type, big.log_focus)
    assert np.array_equal(a[0][1], orc.zbuf()) and np.array_equal(a[0][0], orc.cbuf())


@pytest.mark.parametrize("rt", [S.M_FLAT, S.M_SHADED, S.M_TEXTURED])
def test_clipped_polygons_all_kinds(rt):
    """polygons of 3..6 edges that meet the near plane and the screen borders, every fill;
    the textured case has differing vertex lights (draw-order kernel) and both mappers"""
    a = S.gen_fuzz_polys(900, 320, 240, 8, 900 + rt, render_type=rt, textured=rt == S.M_TEXTURED, zrange=(12, 2000))
    b = S.gen_fuzz_polys(120, 320, 240, 8, 950 + rt, render_type=rt, textured=rt == S.M_TEXTURED, zrange=(12, 2000))
    b.textures = a.textures
    _same(_frames([a, b], True, 320, 240), _frames([a, b], False, 320, 240))


def test_mixed_with_allocating_submits_in_one_frame():
    """no-alloc and allocating submits into the same frame, in draw order"""
    import torch  # noqa: F401
    from raytracing_amd import gpu as G
    a = S.gen_c2(5000, 400, 300, 8, radius=30, seed=21)
    b = S.gen_c2(3000, 400, 300, 8, radius=30, seed=22)
    c3 = S.gen_c2(2000, 400, 300, 8, radius=30, seed=23)
    res = []
    for modes in ((False, True, False), (False, False, False)):
        ctx = G.Context(400, 300)
        ctx.clear(0)
        for sc, na in zip((a, b, c3), modes):
            (v, po, pv), aux = _dev(sc)
            f = ctx.submit_object_dev_noalloc if na else ctx.submit_object_dev
            f(v, po, pv, sc.render_type, sc.log_focus, matrix=sc.matrix, max_edges=3)
        col, z = ctx.readback()
        res.append((col, z, ctx.counters()))
        ctx.close()
    _same(res[:1], res[1:])


def test_two_contexts_share_the_array():
    """two contexts on one GPU take turns with the static array; frames do not mix"""
    from raytracing_amd import gpu as G
    a = S.gen_c2(8000, 320, 240, 8, radius=12, seed=31)
    b = S.gen_c2(6000, 320, 240, 8, radius=12, seed=32)
    want = _frames([a], False, 320, 240) + _frames([b], False, 320, 240)
    c1, c2 = G.Context(320, 240), G.Context(320, 240)
    got = []
    (va, pa, pva), _ = _dev(a)
    (vb, pb, pvb), _ = _dev(b)
    c1.clear(0x102030)
    c2.clear(0x102030)
    c1.submit_object_dev_noalloc(va, pa, pva, 4, a.log_focus, matrix=a.matrix, max_edges=3)
    c2.submit_object_dev_noalloc(vb, pb, pvb, 4, b.log_focus, matrix=b.matrix, max_edges=3)
    for ctx in (c1, c2):
        col, z = ctx.readback()
        got.append((col, z, dict(ctx.counters(), flush_code=0)))
        ctx.close()
    _same(got, want)


def test_too_large_is_refused_and_draws_nothing():
    from raytracing_amd import gpu as G
    sc = S.gen_c2(200000, 320, 240, 8, radius=4, seed=41)
    ctx = G.Context(320, 240)
    ctx.clear(0x00aa00)
    (v, po, pv), _ = _dev(sc)
    with pytest.raises(G.LtError) as e:
        ctx.submit_object_dev_noalloc(v, po, pv, 4, sc.log_focus, matrix=sc.matrix, max_edges=3)
    assert e.value.code == -4 and "static array holds" in str(e.value)
    c, z = ctx.readback()
    assert (c == 0x00aa00).all() and (z == 2 ** 31 - 1).all()
    assert ctx.counters()["polys_in"] == 0
    ctx.close()
