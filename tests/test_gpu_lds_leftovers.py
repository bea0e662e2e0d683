"""What a workgroup finds in LDS words it has not written is whatever the last workgroup on
that CU left there.  The fill's coarse depth table (tiles with long lists) is refreshed by
every thread for its own entry with no barrier before other waves read it; entries must not
rule anything out before their first refresh.  The test leaves the worst value in every CU's
LDS before each frame (a test-only kernel) and compares frames of long tile lists with the
CPU restatement."""
import ctypes as C
import os

import numpy as np
import pytest

from raytracing_amd import scenes as S

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def env():
    import torch
    import oracle.port as P
    from raytracing_amd import gpu as G
    so = os.path.join(HERE, "liblt_kat.so")
    assert os.path.exists(so), "tests/liblt_kat.so not built (run __graft_entry__.build())"
    L = C.CDLL(so)
    L.lt_kat_fill_lds.argtypes = [C.c_int, C.c_void_p]
    L.lt_kat_fill_lds.restype = C.c_int
    return torch, P, G, L


@pytest.mark.parametrize("rt,textured", [(S.M_SHADED, False), (S.M_TEXTURED, True)])
def test_long_tile_lists_after_leftover_lds(env, rt, textured):
    torch, P, G, L = env
    W, H = 256, 128
    # a few tiles, thousands of polygons each: the coarse table is in use in every tile
    sc = S.gen_c3(40000, W, H, 7, radius=14, equal_lights=True, lin_percent=100, seed=0xC0FFEEA0) if textured else \
        S.gen_c2(40000, W, H, 7, radius=14, seed=0xC0FFEEA1)
    orc = P.Oracle(W, H)
    for h, (tex, lg) in enumerate(sc.textures):
        orc.set_texture(h, tex, lg)
    orc.clear(0, 0, 0, 0)
    orc.render_polys(P.transform(sc.matrix, sc.verts), sc.polys, sc.polyaux, sc.polyverts, sc.render_type, sc.log_focus, sc.force_linear)
    sink = torch.zeros(1, dtype=torch.int32, device="cuda")
    ctx = G.Context(W, H)
    for value in (-2**31, 0, -2**31, 1, -2**31, -2**31):
        assert L.lt_kat_fill_lds(value, C.c_void_p(sink.data_ptr())) == 0
        ctx.clear(0)
        ctx.submit_scene(sc)
        c, z = ctx.readback(check=False)
        assert np.array_equal(z, orc.zbuf()) and np.array_equal(c, orc.cbuf()), value
    assert ctx.counters()["tile_refs"] // 6 >= 1024 * (W // 64) * (H // 32) // 2     # lists long enough for the coarse table
    ctx.close()
