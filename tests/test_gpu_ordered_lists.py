"""The draw-order fill (textured polygons whose light varies along spans: the reference
advances light only on depth-test passes, GraphicsPoly.cpp:485-493, 666-674) with per-band
polygon lists, against the CPU restatement and against the variant that reads every header:
small polygons (one span per lane, drawn in rounds), large and clipped polygons (span sweeps),
both texture mappers, several batches in one frame, row bands, list overflow."""
import numpy as np
import pytest

from raytracing_amd import bands as B, scenes as S

pytestmark = pytest.mark.gpu
LISTS_ALWAYS, LISTS_NEVER = 4, 8


@pytest.fixture(scope="module")
def env():
    import oracle.port as P
    from raytracing_amd import gpu as G
    return P, G


def _oracle(P, scenes, w, h, clear=(0x102030, 255, 255, 255)):
    orc = P.Oracle(w, h)
    orc.clear(*clear)
    orc.reset_counters()
    for sc in scenes:
        for hnd, (tex, lg) in enumerate(sc.textures):
            orc.set_texture(hnd, tex, lg)
        orc.render_polys(P.transform(sc.matrix, sc.verts), sc.polys, sc.polyaux, sc.polyverts, sc.render_type, sc.log_focus, sc.force_linear)
    return orc


def _gpu(P, G, scenes, w, h, mode, clear=(0x102030, 255, 255, 255), band=None, capacity=0):
    ctx = G.Context(w, h)
    ctx.set_textured_order(mode)
    if capacity:
        ctx.set_list_capacity(capacity)
    if band:
        ctx.set_band(*band)
    ctx.clear(P.cl_light(*clear))
    ctx.reset_counters()
    for sc in scenes:
        ctx.submit_scene(sc)
    return ctx


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_polygons_of_all_sizes(env, seed):
    """3..6-gon fuzz: near-plane and border clipping, spans from one pixel to the whole row"""
    P, G = env
    sc = S.gen_fuzz_polys(150, 320, 200, 8, 7700 + seed, render_type=S.M_TEXTURED, textured=True, equal_lights=False, zrange=(12, 2000))
    orc = _oracle(P, [sc], 320, 200)
    ctx = _gpu(P, G, [sc], 320, 200, LISTS_ALWAYS)
    c, z = ctx.readback(check=False)
    assert (z == orc.zbuf()).all() and (c == orc.cbuf()).all()
    assert ctx.counters()["fragments"] == orc.counters()["fragments"]


@pytest.mark.parametrize("lin_percent", [100, 70])
def test_small_triangles_1080p_lists_and_headers_agree(env, lin_percent):
    """60k small textured triangles with random vertex lights at 1080p, some through the
    perspective mapper: the restatement, the list-driven fill and the every-header fill"""
    P, G = env
    sc = S.gen_c3(60000, 1920, 1080, 10, radius=10, equal_lights=False, lin_percent=lin_percent, seed=0xC0FFEE53, regular_mapping=True)
    orc = _oracle(P, [sc], 1920, 1080)
    frames = []
    for mode in (LISTS_ALWAYS, LISTS_NEVER, 0):
        ctx = _gpu(P, G, [sc], 1920, 1080, mode)
        c, z = ctx.readback(check=False)
        frames.append((c, z, ctx.counters()["fragments"]))
        ctx.close()
    for c, z, fr in frames:
        assert (z == orc.zbuf()).all() and (c == orc.cbuf()).all()
        assert fr == orc.counters()["fragments"]


def test_dense_overlap_many_rounds(env):
    """thousands of tiny triangles piled on a few cells: long chains of spans that must wait
    for each other inside one group of 64"""
    P, G = env
    sc = S.gen_c3(30000, 256, 96, 7, radius=5, equal_lights=False, lin_percent=100, seed=0xC0FFEE54)
    orc = _oracle(P, [sc], 256, 96)
    ctx = _gpu(P, G, [sc], 256, 96, LISTS_ALWAYS)
    c, z = ctx.readback(check=False)
    assert (z == orc.zbuf()).all() and (c == orc.cbuf()).all()


@pytest.mark.parametrize("radius,lin_percent", [(150, 100), (60, 70), (400, 100)])
def test_groups_of_long_spans(env, radius, lin_percent):
    """wide triangles only: every group of 64 list entries is all long spans, more chunks with
    passes and more sequence numbers than a group records, so the rest of a group is drawn with
    both turns held"""
    P, G = env
    sc = S.gen_c3(2500, 640, 480, 9, radius=radius, equal_lights=False, lin_percent=lin_percent, seed=0xC0FFEE90 + radius,
                  regular_mapping=True)
    orc = _oracle(P, [sc], 640, 480)
    ctx = _gpu(P, G, [sc], 640, 480, LISTS_ALWAYS)
    c, z = ctx.readback(check=False)
    assert (z == orc.zbuf()).all() and (c == orc.cbuf()).all()
    assert ctx.counters()["fragments"] == orc.counters()["fragments"]


def test_several_batches_in_one_frame_and_frames_in_flight(env):
    """three textured batches drawn over each other, twice without a flush (two frames in
    flight, work sets in turns): every batch has its own lists"""
    P, G = env
    scs = [S.gen_c3(25000, 640, 480, 9, radius=12, equal_lights=False, lin_percent=100, seed=0xC0FFEE60 + k) for k in range(3)]
    for sc in scs[1:]:
        sc.textures = scs[0].textures
    orc = _oracle(P, scs, 640, 480)
    ctx = _gpu(P, G, scs, 640, 480, LISTS_ALWAYS)
    ctx.clear(P.cl_light(0x102030, 255, 255, 255))
    for sc in scs:
        ctx.submit_scene(sc)
    c, z = ctx.readback(check=False)
    assert (z == orc.zbuf()).all() and (c == orc.cbuf()).all()
    assert ctx.counters()["fragments"] == 2 * orc.counters()["fragments"]


def test_row_bands(env):
    """a context that draws only its band of rows (sort-first split) with lists"""
    P, G = env
    sc = S.gen_c3(40000, 800, 600, 9, radius=14, equal_lights=False, lin_percent=100, seed=0xC0FFEE70)
    orc = _oracle(P, [sc], 800, 600)
    c = np.empty((600, 800), np.int32)
    z = np.empty((600, 800), np.int32)
    frags = 0
    for y0, y1 in B.all_bands(3, 600):
        ctx = _gpu(P, G, [sc], 800, 600, LISTS_ALWAYS, band=(y0, y1))
        cb, zb = ctx.readback(check=False)
        c[y0:y1], z[y0:y1] = cb[y0:y1], zb[y0:y1]
        frags += ctx.counters()["fragments"]
        ctx.close()
    assert (z == orc.zbuf()).all() and (c == orc.cbuf()).all()
    assert frags == orc.counters()["fragments"]


def test_list_overflow_is_reported(env):
    P, G = env
    sc = S.gen_c3(30000, 640, 480, 9, radius=12, equal_lights=False, lin_percent=100, seed=0xC0FFEE80)
    ctx = _gpu(P, G, [sc], 640, 480, LISTS_ALWAYS, capacity=4000)
    with pytest.raises(G.LtError) as e:
        ctx.flush()
    assert e.value.code == -4


def _wide_quad(width, y0, y1, z, l0, l1):
    t = np.zeros((5, 8), np.int32)
    for k, (x, y) in enumerate([(0, y0), (width - 1, y0), (width - 1, y1), (0, y1)]):
        li = l0 if x == 0 else l1
        t[k] = (x, y, z, li << 16, (li // 2) << 16, (255 - li) << 16, (x * 3) << 16, (y * 5) << 16)
    t[4] = t[0]
    return (4, t, 0, 0)


@pytest.mark.parametrize("nquads", [5, 6, 7, 8, 13])
def test_chunk_records_overflow_inside_a_span(env, nquads):
    """full-width quads, each nearer than the one before: ten 64-pixel chunks with passes a span.  With seven
    quads the 64 chunk records of the group run out inside the nearest quad's span, and what was recorded of
    that span must still be written (nothing covers it afterwards)"""
    P, G = env
    W, H = 640, 64
    tex = S.make_texture(9, 6)
    polys = [_wide_quad(W, 8, 40, -1000 - 1000 * k, 40 + 10 * k, 250 - 10 * k) for k in range(nquads)]
    orc = P.Oracle(W, H)
    orc.set_texture(0, tex, 6)
    orc.clear(0, 0, 0, 0)
    for (e, t, col, tx) in polys:
        orc.g_polygon(8, t, colour=col, tex=0)
    for mode in (1 | LISTS_ALWAYS, 1 | LISTS_NEVER):
        ctx = G.Context(W, H)
        ctx.texture_upload(0, tex, 6)
        ctx.set_textured_order(mode)
        ctx.clear(0)
        ctx.submit_screen_polys(8, polys)
        c, z = ctx.readback(check=False)
        assert (z == orc.zbuf()).all() and (c == orc.cbuf()).all(), mode
        ctx.close()


def test_prefixes_of_scenes_of_wide_polygons(env):
    """the first k polygons of a scene, for many k: what a polygon writes is compared before later
    polygons cover it (a whole scene of wide polygons hides most of its own draws)"""
    P, G = env
    W, H = 640, 96
    for seed, radius, lin in ((6, 500, 90), (2, 400, 100), (3, 60, 70)):
        sc = S.gen_c3(400, W, H, 8, radius=radius, equal_lights=False, lin_percent=lin, seed=0xC0FFEE00 + seed, regular_mapping=True)
        ctx = G.Context(W, H)
        orc = P.Oracle(W, H)
        for h, (tex, lg) in enumerate(sc.textures):
            ctx.texture_upload(h, tex, lg)
            orc.set_texture(h, tex, lg)
        ctx.set_textured_order(1 | LISTS_ALWAYS)
        vv = P.transform(sc.matrix, sc.verts)
        for k in list(range(1, 100, 3)) + list(range(100, 400, 11)):
            orc.clear(0, 0, 0, 0)
            orc.render_polys(vv, sc.polys[:k], sc.polyaux[:k], sc.polyverts, sc.render_type, sc.log_focus, sc.force_linear)
            ctx.clear(0)
            ctx.submit_object(sc.verts, sc.polys[:k], sc.polyverts, sc.render_type, sc.log_focus, polyaux=sc.polyaux[:k],
                              matrix=sc.matrix, force_linear=sc.force_linear, max_edges=3)
            c, z = ctx.readback(check=False)
            assert (z == orc.zbuf()).all() and (c == orc.cbuf()).all(), (seed, k)
        ctx.close()


def test_piles_of_overlapping_triangles(env):
    """the same few pixels drawn over and over (every span of a group waits for the one before: more rounds
    than a group numbers, so it ends with both turns held), now and then a wide triangle between them;
    prefixes of the list, so that a draw is seen before the next one covers it"""
    P, G = env
    W, H = 128, 64
    tex = S.make_texture(9, 6)
    rng = np.random.default_rng(5)
    for variant in (0, 4):
        polys = []
        for k in range(220):
            cx, cy = 40 + (k % 3) * (variant % 3), 30 + (k % 2)
            r = 5 if (k % 17 or variant < 3) else 60
            t = np.zeros((4, 8), np.int32)
            for j, (x, y) in enumerate([(cx - r, cy - r), (cx + r, cy), (cx - r // 2, cy + r)]):
                t[j, :3] = (x, y, int(rng.integers(-50000, -100)) if j == 0 else t[0, 2])
                t[j, 3:6] = rng.integers(0, 256, 3) << 16
                t[j, 6:8] = rng.integers(0, 1000, 2) << 16
            t[3] = t[0]
            polys.append((3, t, 0, 0))
        for mode in (1 | LISTS_ALWAYS, 1 | LISTS_NEVER):
            ctx = G.Context(W, H)
            ctx.texture_upload(0, tex, 6)
            ctx.set_textured_order(mode)
            for k in list(range(60, 221, 7)) + [220]:
                orc = P.Oracle(W, H)
                orc.set_texture(0, tex, 6)
                orc.clear(0, 0, 0, 0)
                for (e, t, col, tx) in polys[:k]:
                    orc.g_polygon(8, t, colour=col, tex=0)
                ctx.clear(0)
                ctx.submit_screen_polys(8, polys[:k])
                c, z = ctx.readback(check=False)
                assert (z == orc.zbuf()).all() and (c == orc.cbuf()).all(), (variant, mode, k)
            ctx.close()
