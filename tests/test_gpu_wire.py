"""Wireframe drawing on the GPU (LT_WIRE objects, lt_submit_lines, the mirror's G_line) against
the CPU restatement and the digests of frames the reference build drew, bit for bit."""
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
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "wire_kats.json")))
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
    steps = MG.wire_cases()[name]
    orc = P.Oracle(W, H)
    oc, oz = MG.run_wire_case(steps, orc)
    ctx = G.Context(W, H)
    ctx.set_pipelining(pipelined)
    c, z = run_gpu(P, ctx, steps, device_arrays=pipelined)
    assert S.frame_digest(c) == GOLD[name]["colour"] and S.frame_digest(z) == GOLD[name]["depth"]
    assert np.array_equal(c, oc) and np.array_equal(z, oz)
    k, ok = ctx.counters(), orc.counters()
    assert k["fragments"] == ok["fragments"] and k["polys_drawn"] == ok["polys_drawn"]


def test_wire_object_near_plane_and_limits(env):
    """a larger object across the near plane at 640x480; a polygon with a bad vertex index is
    reported and the rest is drawn"""
    P, G = env
    sc = S.gen_fuzz_polys(3000, 640, 480, 9, 5300, render_type=S.M_FLAT)
    m = P.make_matrix((3, 250, 1, 0, 0, 0), (0, 9, 0, 5, 5, 40), 0)
    orc = P.Oracle(640, 480)
    orc.clear(0, 0, 0, 0)
    orc.render_polys(P.transform(m, sc.verts), sc.polys, sc.polyaux, sc.polyverts, S.M_WIRE, sc.log_focus)
    ctx = G.Context(640, 480)
    ctx.clear(0)
    ctx.submit_object(sc.verts, sc.polys, sc.polyverts, S.M_WIRE, sc.log_focus, polyaux=sc.polyaux, matrix=m)
    c, z = ctx.readback()
    assert np.array_equal(c, orc.cbuf()) and np.array_equal(z, orc.zbuf())
    assert ctx.counters()["fragments"] == orc.counters()["fragments"] > 10000
    pv = sc.polyverts.copy()
    pv[sc.polys[7, 3], 0] = sc.verts.shape[0] + 5
    ctx.clear(0)
    ctx.submit_object(sc.verts, sc.polys, pv, S.M_WIRE, sc.log_focus, polyaux=sc.polyaux, matrix=m)
    with pytest.raises(G.LtError):
        ctx.flush()
    assert ctx.counters()["overflow_polys"] == 1


def test_bands_add_up(env):
    """lines drawn band by band into one frame equal the frame drawn at once"""
    import torch
    P, G = env
    steps = MG.wire_cases()["mixed_unclipped"]
    whole = run_gpu(P, G.Context(W, H), steps)
    f_c = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    f_z = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    frags = 0
    for (y0, y1) in [(0, 64), (64, 128), (128, H)]:
        ctx = G.Context(W, H)
        ctx.bind_framebuffer(f_c, f_z)
        ctx.set_band(y0, y1)
        run_gpu(P, ctx, steps)
        frags += ctx.counters()["fragments"]
    assert np.array_equal(f_c.cpu().numpy(), whole[0]) and np.array_equal(f_z.cpu().numpy(), whole[1])
    ctx = G.Context(W, H)
    run_gpu(P, ctx, steps)
    assert frags == ctx.counters()["fragments"]


def test_mirror_g_line_and_wire_object(env):
    """G_line and M_render_polygon_object with M_WIRE through the C++ mirror of the reference's API"""
    import torch  # noqa: F401
    from raytracing_amd import _lib
    P, G = env
    _lib.lib()
    eng = C.CDLL(os.path.join(os.path.dirname(_lib.SO_PATH), "liblt_engine.so"))
    eng.lte_cbuf.restype = C.POINTER(C.c_int32)
    eng.lte_zbuf.restype = C.POINTER(C.c_int32)
    eng.lte_init(W, H)
    sc = S.gen_fuzz_polys(120, W, H, 7, 5400, render_type=S.M_FLAT)
    obj = (7, 9, 250, 3, 4, 60)
    lines = MG.wire_lines(91, 150, -60, 60, -500, 500)
    orc = P.Oracle(W, H)
    orc.clear(0x203040, 250, 240, 230)
    m = P.make_matrix((0, 0, 0, 0, 0, 0), obj, 0)
    orc.render_polys(P.transform(m, sc.verts), sc.polys, sc.polyaux, sc.polyverts, S.M_WIRE, sc.log_focus)
    for l in lines:
        orc.g_line(l[0:3], l[3:6], int(l[6]), 200, 100, 255)
    eng.lte_clear(0x203040, 250, 240, 230)
    v, po, pa, pv = sc.verts.copy(), sc.polys.copy(), sc.polyaux.copy(), sc.polyverts.copy()
    eng.lte_render_object(_p(v), v.shape[0], _p(po), _p(pa), _p(pv), po.shape[0], S.M_WIRE, 0, 0, 0, 0, 0, 0, sc.log_focus,
                          *obj[3:], *obj[:3])
    for l in lines:
        a, b = np.ascontiguousarray(l[0:3]), np.ascontiguousarray(l[3:6])
        eng.lte_g_line(_p(a), _p(b), int(l[6]), 200, 100, 255)
    for k, l in enumerate(MG.wire_lines(92, 200, -5, 5, -500, 500)):      # G_dot and G_pixel in turn, over the lines
        a = np.ascontiguousarray(l[0:3])
        if k % 2:
            orc.g_dot(a, int(l[6]), 90, 255, 30)
            eng.lte_g_dot(_p(a), int(l[6]), 90, 255, 30)
        elif 0 <= a[0] < W and 0 <= a[1] < H:
            orc.g_pixel(a, int(l[6]))
            eng.lte_g_pixel(_p(a), int(l[6]))
    eng.lte_flush()
    c = np.ctypeslib.as_array(eng.lte_cbuf(), shape=(H, W)).copy()
    z = np.ctypeslib.as_array(eng.lte_zbuf(), shape=(H, W)).copy()
    assert np.array_equal(c, orc.cbuf()) and np.array_equal(z, orc.zbuf())
