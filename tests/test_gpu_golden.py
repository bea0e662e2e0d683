"""HIP path against the committed vectors that the reference's own code produced."""
import fnmatch

import numpy as np
import pytest

from raytracing_amd import scenes as S
from tests import golden_util as GU

pytestmark = pytest.mark.gpu

FRAMES = GU.frames()
# scenes that need parts of the reference this library does not draw yet: they must be
# refused with LT_ERR_UNSUPPORTED, never drawn approximately without saying so
NEEDS_PRP_OR_ORDER = ["fuzz160x120-rt8-*", "c3q2-*"]


def _unsupported(name):
    return any(fnmatch.fnmatch(name, p) for p in NEEDS_PRP_OR_ORDER)


@pytest.mark.parametrize("name", sorted(FRAMES))
def test_whole_frames(name):
    import oracle.port as P
    from raytracing_amd import gpu as G
    g = FRAMES[name]
    sc = GU.build_scene(g["spec"])
    ctx = G.Context(sc.width, sc.height)
    ctx.clear(P.cl_light(*g["clear"]))
    ctx.submit_scene(sc)
    if _unsupported(name):
        with pytest.raises(G.LtError) as ei:
            ctx.readback()
        assert ei.value.code == -5
        return
    c, z = ctx.readback()
    assert S.frame_digest(z) == g["depth"]
    assert S.frame_digest(c) == g["colour"]
    cr = GU.crops()
    if name + "_c" in cr:
        assert np.array_equal(c[200:232, 300:332], cr[name + "_c"])


def test_configuration_1_tracer_world():
    from raytracing_amd import gpu as G
    g = GU.c1()
    W, H, LF = [int(v) for v in g["meta"]]
    ctx = G.Context(W, H)
    ctx.clear(0)
    for i in range(int(g["n_objs"][0])):
        ctx.submit_object(g[f"obj{i}_verts"], g[f"obj{i}_polys"], g[f"obj{i}_polyverts"], G.LT_SHADED, LF,
                          matrix=g[f"obj{i}_m12"], max_edges=4)
    c, z = ctx.readback()
    assert np.array_equal(z, g["depth"]) and np.array_equal(c, g["colour"])


def test_library_is_the_hip_one():
    """The frame above came from the in-tree HIP library, on a GPU."""
    import torch
    from raytracing_amd import _lib
    assert torch.cuda.is_available()
    assert _lib.lib() is not None
    maps = open("/proc/self/maps").read()
    assert "raytracing_amd/libltgpu.so" in maps
