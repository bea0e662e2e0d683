"""Host arrays of every size class reach the device unchanged: arrays below the staging threshold, arrays of one
piece of the pinned ring, arrays of several pieces with a ragged end, and arrays that are pinned memory already;
each frame against the same arrays submitted from device memory."""
import numpy as np
import pytest

from raytracing_amd import scenes as S

pytestmark = pytest.mark.gpu


def arrays(sc):
    aux = sc.polyaux if sc.polyaux is not None else np.zeros((sc.n_polys, 4), np.int32)
    return [np.ascontiguousarray(a, dtype=np.int32) for a in (sc.verts, sc.polys, sc.polyverts, aux)]


def frame_of(G, sc, how):
    import torch
    ctx = G.Context(sc.width, sc.height)
    ctx.clear(0x203040)
    me = 3
    if how == "device":
        d = [torch.from_numpy(a).cuda() for a in arrays(sc)]
        ctx.submit_object_dev(d[0], d[1], d[2], sc.render_type, sc.log_focus, d_polyaux=d[3], matrix=sc.matrix, max_edges=me)
    elif how == "pinned":
        keep = [torch.from_numpy(a).pin_memory() for a in arrays(sc)]
        v, po, pv, pa = [t.numpy() for t in keep]
        ctx.submit_object(v, po, pv, sc.render_type, sc.log_focus, polyaux=pa, matrix=sc.matrix, max_edges=me)
    else:
        v, po, pv, pa = arrays(sc)
        ctx.submit_object(v, po, pv, sc.render_type, sc.log_focus, polyaux=pa, matrix=sc.matrix, max_edges=me)
    c, z = ctx.readback()
    k = ctx.counters()
    ctx.close()
    return c, z, k["fragments"]


@pytest.mark.parametrize("n", [3000, 61003, 400001])
def test_host_arrays_of_every_size_class(n):
    """3000 triangles: every array below 1 MB (HIP's own pageable copy); 61003: pieces of up to 6 MB; 400001: the
    vertex records are 38 MB, three pieces with a ragged end, the vertices 19 MB"""
    from raytracing_amd import gpu as G
    sc = S.gen_c2(n, 1280, 720, 9, radius=6, seed=n)
    ref = frame_of(G, sc, "device")
    for how in ("host", "pinned", "host"):
        c, z, f = frame_of(G, sc, how)
        assert np.array_equal(c, ref[0]) and np.array_equal(z, ref[1]) and f == ref[2], how


def test_consecutive_host_submits_reuse_the_ring():
    """two different objects from host arrays into one frame, back to back: the second call's pieces go into the
    ring while the first call's batch is still being set up"""
    from raytracing_amd import gpu as G
    import torch
    a = S.gen_c2(150000, 1280, 720, 9, radius=6, seed=5)
    b = S.gen_c2(90001, 1280, 720, 9, radius=10, seed=6)
    frames = []
    for how in ("device", "host"):
        ctx = G.Context(1280, 720)
        ctx.clear(0x203040)
        for sc in (a, b, a):
            if how == "device":
                d = [torch.from_numpy(x).cuda() for x in arrays(sc)]
                ctx.submit_object_dev(d[0], d[1], d[2], sc.render_type, sc.log_focus, d_polyaux=d[3], matrix=sc.matrix, max_edges=3)
            else:
                v, po, pv, pa = arrays(sc)
                ctx.submit_object(v, po, pv, sc.render_type, sc.log_focus, polyaux=pa, matrix=sc.matrix, max_edges=3)
        frames.append(ctx.readback())
        ctx.close()
    assert np.array_equal(frames[0][0], frames[1][0]) and np.array_equal(frames[0][1], frames[1][1])


@pytest.mark.parametrize("threads", [1, 2, 7, 16])
def test_frame_is_independent_of_copy_threads(threads):
    """lt_set_copy_threads: the pieces of the ring are filled by that many host threads; changed between two
    submits of one context as well"""
    from raytracing_amd import gpu as G
    sc = S.gen_c2(200003, 1280, 720, 9, radius=6, seed=12)
    ref = frame_of(G, sc, "device")
    ctx = G.Context(sc.width, sc.height)
    with pytest.raises(G.LtError):
        ctx.set_copy_threads(0)
    with pytest.raises(G.LtError):
        ctx.set_copy_threads(17)
    v, po, pv, pa = arrays(sc)
    for t in (threads, 4, threads):
        ctx.set_copy_threads(t)
        ctx.clear(0x203040)
        ctx.reset_counters()
        ctx.submit_object(v, po, pv, sc.render_type, sc.log_focus, polyaux=pa, matrix=sc.matrix, max_edges=3)
        c, z = ctx.readback()
        assert np.array_equal(c, ref[0]) and np.array_equal(z, ref[1]) and ctx.counters()["fragments"] == ref[2]
    ctx.close()


def test_arrays_in_the_librarys_pinned_memory():
    """lt_host_alloc / lt_host_free: arrays kept there go to the copy engine as they are"""
    from raytracing_amd import gpu as G
    sc = S.gen_c2(120001, 1280, 720, 9, radius=6, seed=13)
    ref = frame_of(G, sc, "device")
    held = []
    for a in arrays(sc):
        h = G.host_alloc(a.shape)
        h[...] = a
        held.append(h)
    ctx = G.Context(sc.width, sc.height)
    ctx.clear(0x203040)
    ctx.submit_object(held[0], held[1], held[2], sc.render_type, sc.log_focus, polyaux=held[3], matrix=sc.matrix, max_edges=3)
    c, z = ctx.readback()
    assert np.array_equal(c, ref[0]) and np.array_equal(z, ref[1])
    ctx.close()
    for h in held:
        G.host_free(h)
    assert G._lib.lib().lt_host_free(None) == 0
    p = G.C.c_void_p()
    assert G._lib.lib().lt_host_alloc(0, G.C.byref(p)) != 0
