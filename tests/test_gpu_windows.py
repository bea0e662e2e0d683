"""Clip windows smaller than the screen on the GPU (lt_set_clip_window; the mirror's
C_init_clipping / C_set_bounds) against the CPU restatement and the reference-made digests."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

from raytracing_amd import scenes as S

sys.path.insert(0, os.path.dirname(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden as MG  # noqa: E402
from gpu_steps import run_gpu  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "window_kats.json")))
W, H = MG.WIRE_W, MG.WIRE_H


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def env():
    import oracle.port as P
    from raytracing_amd import gpu as G
    return P, G


@pytest.mark.parametrize("pipelined", [True, False])
@pytest.mark.parametrize("name", sorted(GOLD))
def test_golden_frames(env, name, pipelined):
    P, G = env
    steps = MG.window_cases()[name]
    orc = P.Oracle(W, H)
    for h, (tex, lg) in enumerate(MG._textures_of(steps)):
        orc.set_texture(h, tex, lg)
    oc, oz = MG.run_wire_case(steps, orc)
    ctx = G.Context(W, H)
    ctx.set_pipelining(pipelined)
    c, z = run_gpu(P, ctx, steps, device_arrays=pipelined)
    assert np.array_equal(c, oc) and np.array_equal(z, oz)
    assert S.frame_digest(c) == GOLD[name]["colour"] and S.frame_digest(z) == GOLD[name]["depth"]
    k, ok = ctx.counters(), orc.counters()
    assert k["fragments"] == ok["fragments"]
    if not any(st[0] == "gpolys" for st in steps):              # the checker counts polygons at the M_render_polygon level only
        assert k["polys_drawn"] == ok["polys_drawn"]


def test_triangle_soup_under_a_window(env):
    """the fast set-up path: 200k small triangles at 640x480, window in the middle; then the
    window is widened for a second object in the same frame"""
    import torch
    P, G = env
    a = S.gen_c2(200000, 640, 480, 9, radius=8, seed=81)
    b = S.gen_c2(50000, 640, 480, 9, radius=30, seed=82)
    orc = P.Oracle(640, 480)
    orc.clear(0, 0, 0, 0)
    ctx = G.Context(640, 480)
    ctx.clear(0)
    for sc, win in ((a, (100, 90, 500, 400)), (b, (0, 200, 639, 300))):
        orc.set_clip_window(*win)
        orc.render_polys(P.transform(sc.matrix, sc.verts), sc.polys, sc.polyaux, sc.polyverts, sc.render_type, sc.log_focus)
        ctx.set_clip_window(*win)
        d = [torch.from_numpy(x).cuda() for x in (sc.verts, sc.polys, sc.polyverts)]
        ctx.submit_object_dev(d[0], d[1], d[2], sc.render_type, sc.log_focus, matrix=sc.matrix, max_edges=3)
        ctx.flush()
    c, z = ctx.readback()
    assert np.array_equal(c, orc.cbuf()) and np.array_equal(z, orc.zbuf())
    assert ctx.counters()["fragments"] == orc.counters()["fragments"]


def test_bad_windows_are_refused(env):
    P, G = env
    ctx = G.Context(W, H)
    for win in ((-1, 0, 10, 10), (0, 0, W, 10), (5, 5, 5, 9), (0, 9, 10, 3), (0, 0, 10, H)):
        with pytest.raises(G.LtError):
            ctx.set_clip_window(*win)
    ctx.set_clip_window(0, 0, W - 1, H - 1)


def test_mirror_clip_functions(env):
    """C_init_clipping / C_set_bounds / C_get_bounds of the mirror: draws queued before a change
    keep their window; C_set_bounds intersects and reports an empty intersection"""
    import torch  # noqa: F401
    from raytracing_amd import _lib
    P, G = env
    _lib.lib()
    eng = C.CDLL(os.path.join(os.path.dirname(_lib.SO_PATH), "liblt_engine.so"))
    eng.lte_cbuf.restype = C.POINTER(C.c_int32)
    eng.lte_zbuf.restype = C.POINTER(C.c_int32)
    eng.lte_init(W, H)
    polys = S.gen_screen_polys(60, W, H, seed=83, dim=6)
    orc = P.Oracle(W, H)
    orc.clear(0, 0, 0, 0)
    eng.lte_clear(0, 0, 0, 0)
    for part, win in ((polys[:30], (20, 10, 150, 100)), (polys[30:], (60, 40, 199, 149))):
        orc.set_clip_window(*win)
        eng.lte_init_clipping(*win)
        for (e, t, col) in part:
            orc.g_polygon(4, t, colour=col)
            tt = np.ascontiguousarray(t, dtype=np.int32)
            eng.lte_g_polygon(4, _p(tt), e, col, 0, 0, 0, 0)
    assert eng.lte_set_bounds(0, 0, 100, 120) == 1               # -> (60, 40, 100, 120)
    orc.set_clip_window(60, 40, 100, 120)
    for (e, t, col) in polys[:20]:
        orc.g_polygon(4, t, colour=col ^ 0x00ff00)
        tt = np.ascontiguousarray(t, dtype=np.int32)
        eng.lte_g_polygon(4, _p(tt), e, col ^ 0x00ff00, 0, 0, 0, 0)
    assert eng.lte_set_bounds(0, 0, 50, 149) == 0                # empty: the caller would draw nothing
    eng.lte_flush()
    c = np.ctypeslib.as_array(eng.lte_cbuf(), shape=(H, W)).copy()
    z = np.ctypeslib.as_array(eng.lte_zbuf(), shape=(H, W)).copy()
    eng.lte_init_clipping(0, 0, W - 1, H - 1)
    assert np.array_equal(c, orc.cbuf()) and np.array_equal(z, orc.zbuf())
