"""The fill's tile width (lt_set_tile_width: 64, 128 or chosen by the library) does not change the
frame or the counters: every kind of batch at both widths against the CPU restatement."""
import numpy as np
import pytest

from raytracing_amd import scenes as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import oracle.port as P
    from raytracing_amd import gpu as G
    return P, G


def oracle_frame(P, sc):
    o = P.Oracle(sc.width, sc.height)
    for h, (tex, lg) in enumerate(sc.textures):
        o.set_texture(h, tex, lg)
    o.clear(0x102030, 255, 255, 255)
    o.render_polys(P.transform(sc.matrix, sc.verts), sc.polys, sc.polyaux, sc.polyverts, sc.render_type, sc.log_focus, sc.force_linear)
    return o.cbuf().copy(), o.zbuf().copy(), o.counters()


SCENES = {
    "small": lambda: S.gen_c2(60000, 640, 480, 9, radius=8, seed=3),
    "large": lambda: S.gen_c2(4000, 640, 480, 9, radius=128, seed=4),
    "odd_screen": lambda: S.gen_c2(20000, 333, 217, 8, radius=40, seed=5),
    "flat_fuzz": lambda: S.gen_fuzz_polys(1500, 400, 300, 8, 71, render_type=S.M_FLAT),
    "textured_fuzz": lambda: S.gen_fuzz_polys(1500, 400, 300, 8, 75, render_type=S.M_TEXTURED, textured=True, equal_lights=True),
    "textured_ordered": lambda: S.gen_fuzz_polys(600, 400, 300, 8, 73, render_type=S.M_TEXTURED, textured=True, equal_lights=False),
}


@pytest.mark.parametrize("width,waves", [(64, 4), (64, 8), (128, 0)])
@pytest.mark.parametrize("name", sorted(SCENES))
def test_frame_is_independent_of_tile_width(env, name, width, waves):
    """... and of the number of waves that share a 64-pixel tile (lt_set_tile_waves)"""
    P, G = env
    sc = SCENES[name]()
    oc, oz, ok = oracle_frame(P, sc)
    ctx = G.Context(sc.width, sc.height)
    ctx.set_tile_width(width)
    ctx.set_tile_waves(waves)
    ctx.clear(P.cl_light(0x102030, 255, 255, 255))
    ctx.submit_scene(sc)
    c, z = ctx.readback()
    assert np.array_equal(c, oc) and np.array_equal(z, oz)
    k = ctx.counters()
    assert k["fragments"] == ok["fragments"] and k["polys_drawn"] == ok["polys_drawn"]


def test_library_switches_and_frames_stay_equal(env):
    """large polygons make the library go to wide tiles at the next flush, small ones back; wide
    screens (more than 8192 narrow tiles: counting inside set-up) included"""
    P, G = env
    big = S.gen_c2(3000, 2048, 1100, 10, radius=128, seed=6)
    small = S.gen_c2(40000, 2048, 1100, 10, radius=6, seed=7)
    ctx = G.Context(2048, 1100)
    with pytest.raises(G.LtError):
        ctx.set_tile_width(96)
    for sc in (big, big, small, small, big):
        oc, oz, _ = oracle_frame(P, sc)
        ctx.clear(P.cl_light(0x102030, 255, 255, 255))
        ctx.submit_scene(sc)
        c, z = ctx.readback()                                    # flushes: the next batch may use the other width
        assert np.array_equal(c, oc) and np.array_equal(z, oz)


def test_crowded_tiles_switch_the_library_to_eight_waves_and_frames_stay_equal(env):
    """an object drawn small puts most of its polygons into a few tiles (lists of thousands where the average is tens):
    after the next flush the library puts eight waves on a tile; evenly spread polygons take it back to four"""
    P, G = env
    from raytracing_amd import frames as F
    far = S.gen_c2(150000, 1920, 1080, 10, radius=8, seed=8)
    far.matrix = F.frame_matrix(0, translation=(0, 0, 9000))        # the soup at five times its distance: a fifth of its size
    even = S.gen_c2(30000, 1920, 1080, 10, radius=8, seed=9)
    ctx = G.Context(1920, 1080)
    with pytest.raises(G.LtError):
        ctx.set_tile_waves(6)
    for sc in (far, far, even, even, far):
        oc, oz, ok = oracle_frame(P, sc)
        ctx.clear(P.cl_light(0x102030, 255, 255, 255))
        ctx.reset_counters()
        ctx.submit_scene(sc)
        c, z = ctx.readback()
        assert np.array_equal(c, oc) and np.array_equal(z, oz)
        assert ctx.counters()["fragments"] == ok["fragments"]


def depth_tie_scene(n, width, height, log_focus, radius, seed):
    """large triangles of constant depth, a few depths only: polygons that lie wholly behind what their tile
    holds, many of them exactly as deep as it (between equal depths the earlier polygon stays)"""
    sc = S.gen_c2(n, width, height, log_focus, radius=radius, seed=seed)
    v = sc.verts.astype(np.int64)
    for p in range(sc.n_polys):
        first = int(sc.polys[p, 3])
        idx = sc.polyverts[first:first + 3, 0]
        znew = max(256, int(v[idx[0], 2]) // 512 * 512)
        for i in idx:
            zold = int(v[i, 2])
            v[i, 0] = v[i, 0] * znew // zold
            v[i, 1] = v[i, 1] * znew // zold
            v[i, 2] = znew
    sc.verts = v.astype(np.int32)
    return sc


@pytest.mark.parametrize("width", [64, 128])
def test_polygons_behind_what_their_tile_holds(env, width):
    """tile lists of thousands of large polygons (the coarse depth tests are on, and with the wide tile the class
    of polygons that lie wholly behind their tile): frame, depth and the count of depth tests against the CPU
    restatement, drawn once onto a cleared frame and once more onto the frame as it then is"""
    P, G = env
    for sc in (depth_tie_scene(20000, 512, 256, 9, 100, 21), S.gen_c2(20000, 512, 256, 9, radius=100, seed=22)):
        o = P.Oracle(sc.width, sc.height)
        o.clear(0x102030, 255, 255, 255)
        o.reset_counters()
        ctx = G.Context(sc.width, sc.height)
        ctx.set_tile_width(width)
        ctx.clear(P.cl_light(0x102030, 255, 255, 255))
        ctx.reset_counters()
        for again in range(2):
            o.render_polys(P.transform(sc.matrix, sc.verts), sc.polys, sc.polyaux, sc.polyverts, sc.render_type, sc.log_focus, sc.force_linear)
            ctx.submit_scene(sc)
            c, z = ctx.readback()
            assert np.array_equal(z, o.zbuf()) and np.array_equal(c, o.cbuf())
            k = ctx.counters()
            assert k["fragments"] == o.counters()["fragments"] and k["polys_drawn"] == o.counters()["polys_drawn"]
            if width == 128:
                assert k["tile_refs"] >= (again + 1) * 1024 * (sc.width // width) * (sc.height // 32)      # long lists: the coarse tests are on
        ctx.close()
