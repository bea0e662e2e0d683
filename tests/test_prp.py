"""G_prp_textured_polygon (Graphics.h:108-113) at the G_* level: the CPU restatement against
frames the reference build drew through its own function (tests/golden/prp_kats.json)."""
import json
import os
import sys

import pytest

from raytracing_amd import scenes as S

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as MG  # noqa: E402

GOLD = json.load(open(os.path.join(HERE, "golden", "prp_kats.json")))


@pytest.mark.parametrize("name", sorted(GOLD))
def test_restatement_matches_reference_frames(name):
    import oracle.port as P
    c, z = MG.run_prp_case(MG.prp_cases()[name], P.Oracle(MG.WIRE_W, MG.WIRE_H))
    assert S.frame_digest(z) == GOLD[name]["depth"] and S.frame_digest(c) == GOLD[name]["colour"]
    assert GOLD[name]["drawn"] > 500


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(GOLD))
def test_gpu_entry_point_matches_reference_frames(name):
    """lt_submit_screen_polys_prp through the C-ABI"""
    import oracle.port as P
    from raytracing_amd import gpu as G
    polys = MG.prp_cases()[name]
    ctx = G.Context(MG.WIRE_W, MG.WIRE_H)
    ctx.texture_upload(0, S.make_texture(9, 6), 6)
    ctx.clear(P.cl_light(0x000020, 255, 255, 255))
    ctx.submit_screen_polys_prp(polys, 7)
    c, z = ctx.readback()
    assert S.frame_digest(z) == GOLD[name]["depth"] and S.frame_digest(c) == GOLD[name]["colour"]
    # split over three submits and mixed with the linear mapper's entry point: draw order is call order
    oc, oz = c, z
    ctx.clear(P.cl_light(0x000020, 255, 255, 255))
    for part in (polys[:20], polys[20:21], polys[21:]):
        ctx.submit_screen_polys_prp(part, 7)
    c, z = ctx.readback()
    assert (c == oc).all() and (z == oz).all()
    ctx.close()


@pytest.mark.gpu
def test_mirror_function_matches_reference_frames():
    """the host mirror's G_prp_textured_polygon, called as the reference's is"""
    import ctypes as C
    import numpy as np
    import torch  # noqa: F401
    from raytracing_amd import _lib
    _lib.lib()
    L = C.CDLL(os.path.join(os.path.dirname(_lib.SO_PATH), "liblt_engine.so"))
    L.lte_cbuf.restype = C.POINTER(C.c_int32)
    L.lte_zbuf.restype = C.POINTER(C.c_int32)
    W, H = MG.WIRE_W, MG.WIRE_H
    L.lte_init(W, H)
    tex = np.ascontiguousarray(S.make_texture(9, 6), dtype=np.int32)

    def p(a):
        return a.ctypes.data_as(C.c_void_p)

    L.lte_set_texture(3, p(tex), 6)
    for name in sorted(GOLD):
        L.lte_clear(0x000020, 255, 255, 255)
        keep = []
        for (e, t, O, u, v, _, ls) in MG.prp_cases()[name]:
            arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in (t, O, u, v)]
            keep.append(arrs)
            L.lte_g_prp_polygon(p(arrs[0]), e, p(arrs[1]), p(arrs[2]), p(arrs[3]), 3, ls, 7)
        L.lte_flush()
        c = np.ctypeslib.as_array(L.lte_cbuf(), shape=(H, W)).copy()
        z = np.ctypeslib.as_array(L.lte_zbuf(), shape=(H, W)).copy()
        assert S.frame_digest(z) == GOLD[name]["depth"] and S.frame_digest(c) == GOLD[name]["colour"], name
