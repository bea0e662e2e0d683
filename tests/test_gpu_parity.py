"""HIP path vs the CPU restatement (oracle/lt_oracle.c), bit for bit, through the C-ABI."""
import numpy as np
import pytest

from raytracing_amd import scenes as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import oracle.port as P
    from raytracing_amd import gpu as G
    return P, G


def _render_both(P, G, sc, ctx=None, orc=None, clear=(0x102030, 255, 255, 255)):
    own = ctx is None
    if own:
        ctx = G.Context(sc.width, sc.height)
        orc = P.Oracle(sc.width, sc.height)
    for h, (tex, lg) in enumerate(sc.textures):
        orc.set_texture(h, tex, lg)
    orc.clear(*clear)
    orc.reset_counters()
    vv = P.transform(sc.matrix, sc.verts)
    orc.render_polys(vv, sc.polys, sc.polyaux, sc.polyverts, sc.render_type, sc.log_focus, sc.force_linear)
    ctx.clear(P.cl_light(*clear))
    ctx.reset_counters()
    ctx.submit_scene(sc)
    c, z = ctx.readback()
    return c, z, orc.cbuf().copy(), orc.zbuf().copy(), ctx.counters(), orc.counters()


@pytest.mark.parametrize("rt", [S.M_SHADED, S.M_FLAT])
@pytest.mark.parametrize("seed", range(8))
def test_fuzz_polys_small_screen(env, rt, seed):
    P, G = env
    sc = S.gen_fuzz_polys(80, 160, 120, 7, 100 + seed * 3 + rt, render_type=rt)
    c, z, oc, oz, gc, ocn = _render_both(P, G, sc)
    assert (z == oz).all(), f"depth differs at {int((z != oz).sum())} pixels"
    assert (c == oc).all(), f"colour differs at {int((c != oc).sum())} pixels"
    assert gc["fragments"] == ocn["fragments"]
    assert gc["polys_drawn"] == ocn["polys_drawn"]


@pytest.mark.parametrize("zrange", [(1100, 3000), (12, 900)], ids=["linear-mapper", "perspective-mapper"])
@pytest.mark.parametrize("seed", range(6))
def test_fuzz_textured_equal_lights(env, seed, zrange):
    """Textured polygons with one light value per polygon: far ones take
    G_lin_textured_polygon, near ones (first clipped vertex z < 1024, EngBase.cpp:491-493)
    G_prp_textured_polygon with MI_tmapping_vectors / T_linear_solve."""
    P, G = env
    sc = S.gen_fuzz_polys(80, 160, 120, 7, 500 + seed, render_type=S.M_TEXTURED, textured=True,
                          equal_lights=True, zrange=zrange)
    c, z, oc, oz, gc, ocn = _render_both(P, G, sc)
    assert ocn["unsupported"] == 0
    assert (z == oz).all() and (c == oc).all()
    assert gc["fragments"] == ocn["fragments"]


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_textured_varying_lights(env, seed):
    """Light gradients along textured spans: the pass-count dependent light of
    GraphicsPoly.cpp:485-493 / 666-674, through the draw-order kernel."""
    P, G = env
    sc = S.gen_fuzz_polys(80, 160, 120, 7, 900 + seed, render_type=S.M_TEXTURED, textured=True,
                          equal_lights=False, zrange=(12, 2000))
    c, z, oc, oz, gc, ocn = _render_both(P, G, sc)
    assert (z == oz).all() and (c == oc).all()
    assert gc["fragments"] == ocn["fragments"]


def test_c3_varying_lights_1080p(env):
    P, G = env
    sc = S.gen_c3(20000, 1920, 1080, 10, radius=12, equal_lights=False, lin_percent=100, seed=0xC0FFEE43)
    c, z, oc, oz, gc, ocn = _render_both(P, G, sc)
    assert (z == oz).all() and (c == oc).all()


def test_c3_clipped_textured_4k_subset(env):
    """30k clipped textured triangles at 4K.  One of them has a singular texture mapping
    (T_linear_solve would divide by zero, TransBs.cpp:505: the reference crashes there);
    both the checker and the HIP path skip it and say so."""
    P, G = env
    sc = S.gen_c3(30000, 3840, 2160, 11, radius=8, equal_lights=True, lin_percent=90)
    ctx = G.Context(sc.width, sc.height)
    orc = P.Oracle(sc.width, sc.height)
    orc.set_texture(0, *sc.textures[0])
    orc.clear(0, 0, 0, 0)
    orc.render_polys(P.transform(sc.matrix, sc.verts), sc.polys, sc.polyaux, sc.polyverts, sc.render_type, sc.log_focus)
    ctx.clear(0)
    ctx.submit_scene(sc)
    with pytest.raises(G.LtError) as ei:
        ctx.flush()
    assert ei.value.code == -5
    c, z = ctx.readback()
    gc, ocn = ctx.counters(), orc.counters()
    assert gc["unsupported_polys"] == ocn["unsupported"] == 1
    assert (z == orc.zbuf()).all() and (c == orc.cbuf()).all()
    assert gc["fragments"] == ocn["fragments"]
    assert gc["polys_drawn"] == ocn["polys_drawn"]      # includes triangles wholly beyond a screen border


def test_offscreen_triangles_count_as_drawn(env):
    """triangles wholly beyond one screen border but inside the 90-degree view volume: the
    clippers drop every edge, nothing is drawn, polys_drawn still counts the front-facing ones"""
    P, G = env
    W, H, lf = 320, 200, 9
    rng = S.Rng(77)
    n = 4000
    sc = S.gen_c2(n, W, H, lf, radius=10, seed=4242)
    # push every triangle past a border: shift view x or y by a multiple of its depth
    v = sc.verts.copy()
    side = rng.ints(0, 3, n).repeat(3)
    z = v[:, 2]
    v[:, 0] = np.where(side == 0, v[:, 0] - ((z * (W + 80)) >> lf), np.where(side == 1, v[:, 0] + ((z * (W + 80)) >> lf), v[:, 0]))
    v[:, 1] = np.where(side == 2, v[:, 1] - ((z * (H + 80)) >> lf), np.where(side == 3, v[:, 1] + ((z * (H + 80)) >> lf), v[:, 1]))
    sc.verts = v
    c, z, oc, oz, gc, ocn = _render_both(P, G, sc)
    assert (z == oz).all() and (c == oc).all()
    assert gc["fragments"] == ocn["fragments"] == 0
    assert gc["polys_drawn"] == ocn["polys_drawn"] > 0


@pytest.mark.parametrize("layout", ["soup", "odd-first", "shuffled-vertices", "shuffled-records", "tail"])
def test_input_layouts_take_the_same_picture(env, layout):
    """The fast set-up kernel reads a wave's records and vertices as one block when they lie
    in sequence (a triangle soup) and gathers them otherwise: every layout of the same
    triangles must give the same frame."""
    P, G = env
    W, H = 640, 480
    n = 64 * 40 + (17 if layout == "tail" else 0)
    sc = S.gen_c2(n, W, H, 9, radius=12, seed=606)
    rng = np.random.default_rng(5)
    if layout == "odd-first":                      # records start at an odd index: no 16-byte alignment
        sc.polyverts = np.concatenate([np.zeros((1, 6), np.int32), sc.polyverts])
        sc.polys = sc.polys.copy()
        sc.polys[:, 3] += 1
    elif layout == "shuffled-vertices":            # same records, vertex array permuted
        perm = rng.permutation(sc.verts.shape[0])
        inv = np.empty_like(perm)
        inv[perm] = np.arange(perm.size)
        sc.verts = sc.verts[perm]
        sc.polyverts = sc.polyverts.copy()
        sc.polyverts[:, 0] = inv[sc.polyverts[:, 0]]
    elif layout == "shuffled-records":             # record groups permuted, polygons point at them
        perm = rng.permutation(n)
        pv = sc.polyverts.reshape(n, 4, 6)
        sc.polyverts = pv[perm].reshape(-1, 6)
        inv = np.empty_like(perm)
        inv[perm] = np.arange(n)
        sc.polys = sc.polys.copy()
        sc.polys[:, 3] = inv * 4
    c, z, oc, oz, gc, ocn = _render_both(P, G, sc)
    assert (z == oz).all() and (c == oc).all()
    assert gc["fragments"] == ocn["fragments"] and gc["polys_drawn"] == ocn["polys_drawn"]


def test_c2_small_subset_1080p(env):
    P, G = env
    sc = S.gen_c2(20000, 1920, 1080, 10, radius=8)
    c, z, oc, oz, gc, ocn = _render_both(P, G, sc)
    assert (z == oz).all() and (c == oc).all()
    assert gc["fragments"] == ocn["fragments"]


def test_c2_large_subset_1080p(env):
    P, G = env
    sc = S.gen_c2(3000, 1920, 1080, 10, radius=128)
    c, z, oc, oz, gc, ocn = _render_both(P, G, sc)
    assert (z == oz).all() and (c == oc).all()
    assert gc["fragments"] == ocn["fragments"]


def test_two_submits_keep_draw_order(env):
    """Equal-depth overlap across two submits: the earlier submit keeps the pixel."""
    P, G = env
    sc = S.gen_c2(4000, 640, 480, 9, radius=40, seed=77)
    half = sc.n_polys // 2
    ctx = G.Context(640, 480)
    orc = P.Oracle(640, 480)
    orc.clear(0, 0, 0, 0)
    vv = P.transform(sc.matrix, sc.verts)
    orc.render_polys(vv, sc.polys, None, sc.polyverts, sc.render_type, sc.log_focus)
    ctx.clear(0)
    ctx.submit_object(sc.verts, sc.polys[:half], sc.polyverts, sc.render_type, sc.log_focus, max_edges=3)
    ctx.submit_object(sc.verts, sc.polys[half:], sc.polyverts, sc.render_type, sc.log_focus, max_edges=3)
    c, z = ctx.readback()
    assert (z == orc.zbuf()).all() and (c == orc.cbuf()).all()


def test_transform_matches(env):
    import torch
    P, G = env
    rng = S.Rng(5)
    v = rng.ints(-40000, 40000, 4 * 5000).reshape(-1, 4).astype(np.int32)
    m = P.make_matrix((3, 5, 7, 10, -20, 30), (11, 13, 17, 100, 200, 300))
    want = P.transform(m, v)
    ctx = G.Context(64, 64)
    d_in = torch.from_numpy(v).cuda()
    d_out = torch.full_like(d_in, 12345)
    ctx.transform(m, d_in, d_out)
    ctx.flush()
    got = d_out.cpu().numpy()
    assert (got[:, :3] == want[:, :3]).all()
    assert (got[:, 3] == 12345).all()          # 4th lane untouched (TransBs.cpp:391)


def test_screen_polys_shaded(env):
    P, G = env
    W, H = 200, 150
    polys = S.gen_screen_polys(120, W, H, seed=9, dim=6)
    ctx = G.Context(W, H)
    orc = P.Oracle(W, H)
    orc.clear(0x334455, 200, 200, 200)
    ctx.clear(P.cl_light(0x334455, 200, 200, 200))
    for (e, t, col) in polys:
        orc.g_polygon(4, t, colour=col)
    ctx.submit_screen_polys(G.LT_SHADED, polys)
    c, z = ctx.readback()
    assert (z == orc.zbuf()).all() and (c == orc.cbuf()).all()


def test_polygons_of_the_first_rows_only(env):
    """polygons whose rows are [0, 1) or [0, 2): the binning knows a polygon by its packed rows
    (first | last << 16; zero means nothing to draw), which is smallest for these"""
    P, G = env
    W, H = 200, 150
    L = 200 << 16
    def tri(x0, x1, xm, ym, z, light):
        t = np.array([[x0, 0, z, light, light, light], [x1, 0, z, light, light, light], [xm, ym, z, light, light, light],
                      [x0, 0, z, light, light, light]], np.int32)
        return (3, t, 0x00ffffff)
    polys = [tri(0, 9, 4, 1, 500, L), tri(9, 0, 4, 1, 400, L // 2),       # both windings: rows [0, 1)
             tri(20, 60, 40, 2, 300, L), tri(60, 20, 40, 2, 200, L // 2),   # rows [0, 2)
             tri(190, 199, 195, 1, 100, L), tri(199, 190, 195, 1, 90, L // 2),
             tri(100, 140, 120, 0, 80, L)]                                  # a single row of points: rows [0, 0), nothing
    ctx = G.Context(W, H)
    orc = P.Oracle(W, H)
    orc.clear(0x112233, 255, 255, 255)
    ctx.clear(P.cl_light(0x112233, 255, 255, 255))
    for (e, t, col) in polys:
        orc.g_polygon(4, t, colour=col)
    ctx.submit_screen_polys(G.LT_SHADED, polys)
    c, z = ctx.readback()
    assert (z == orc.zbuf()).all() and (c == orc.cbuf()).all()
    assert (z[0] != 2 ** 31 - 1).sum() > 40 and (z[2:] == 2 ** 31 - 1).all()
