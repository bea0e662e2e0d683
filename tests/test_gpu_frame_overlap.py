"""Two frames in flight (lt_set_frame_overlap): a cleared frame starts in the other of two
buffer pairs on a fill stream of its own.  What a caller reads back must not depend on it."""
import numpy as np
import pytest

from raytracing_amd import scenes as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import oracle.port as P
    from raytracing_amd import gpu as G
    return P, G


def _oracle_frame(P, W, H, clear, scenes):
    orc = P.Oracle(W, H)
    orc.clear(clear, 255, 255, 255)
    for sc in scenes:
        orc.render_polys(P.transform(sc.matrix, sc.verts), sc.polys, None, sc.polyverts, sc.render_type, sc.log_focus)
    return orc


@pytest.mark.parametrize("overlap", [True, False])
def test_frames_without_a_flush_and_drawing_on_after_a_read(env, overlap):
    """seven cleared frames back to back (odd and even counts of buffer switches on the way),
    read back; then two more batches drawn over the frame that was read, with no clear"""
    import torch
    P, G = env
    W, H = 640, 480
    ctx = G.Context(W, H)
    ctx.set_frame_overlap(overlap)
    scenes = [S.gen_c2(4000 + 700 * f, W, H, 9, radius=10 + f, seed=3100 + f) for f in range(9)]
    dev = [[torch.from_numpy(a).cuda() for a in (sc.verts, sc.polys, sc.polyverts)] for sc in scenes]

    def draw(f):
        sc, d = scenes[f], dev[f]
        ctx.submit_object_dev(d[0], d[1], d[2], sc.render_type, sc.log_focus, matrix=sc.matrix, max_edges=3)

    frags = 0
    for f in range(7):
        ctx.clear(f * 0x010203)
        draw(f)
        frags += _oracle_frame(P, W, H, 0, [scenes[f]]).counters()["fragments"]
        if f == 3:                                   # a read in the middle: the fourth frame, from whichever pair holds it
            c, z = ctx.readback()
            orc = _oracle_frame(P, W, H, 3 * 0x010203, [scenes[3]])
            assert (z == orc.zbuf()).all() and (c == orc.cbuf()).all()
    c, z = ctx.readback()
    orc = _oracle_frame(P, W, H, 6 * 0x010203, [scenes[6]])
    assert (z == orc.zbuf()).all() and (c == orc.cbuf()).all()
    assert ctx.counters()["fragments"] == frags
    draw(7)
    draw(8)
    c, z = ctx.readback()
    orc = _oracle_frame(P, W, H, 6 * 0x010203, [scenes[6], scenes[7], scenes[8]])
    assert (z == orc.zbuf()).all() and (c == orc.cbuf()).all()


def test_clear_twice_and_empty_frames(env):
    """clears with nothing drawn between them, a frame that is only cleared, then a drawn one"""
    import torch
    P, G = env
    W, H = 320, 200
    ctx = G.Context(W, H)
    sc = S.gen_c2(3000, W, H, 8, radius=9, seed=3200)
    d = [torch.from_numpy(a).cuda() for a in (sc.verts, sc.polys, sc.polyverts)]
    ctx.clear(0x111111)
    ctx.submit_object_dev(d[0], d[1], d[2], sc.render_type, sc.log_focus, matrix=sc.matrix, max_edges=3)
    ctx.clear(0x222222)
    ctx.clear(0x333333)
    c, z = ctx.readback()
    assert (c == 0x333333).all() and (z == 2 ** 31 - 1).all()
    ctx.submit_object_dev(d[0], d[1], d[2], sc.render_type, sc.log_focus, matrix=sc.matrix, max_edges=3)
    c, z = ctx.readback()
    orc = _oracle_frame(P, W, H, 0x333333, [sc])
    assert (z == orc.zbuf()).all() and (c == orc.cbuf()).all()


def test_bound_frame_and_back(env):
    """own frame drawn (in the second pair after two clears), a caller's frame bound and drawn,
    unbound: the own frame is as it was left, and drawing goes on in it"""
    import torch
    P, G = env
    W, H = 320, 200
    ctx = G.Context(W, H)
    a, b = S.gen_c2(2500, W, H, 8, radius=9, seed=3300), S.gen_c2(2500, W, H, 8, radius=9, seed=3301)
    da = [torch.from_numpy(x).cuda() for x in (a.verts, a.polys, a.polyverts)]
    db = [torch.from_numpy(x).cuda() for x in (b.verts, b.polys, b.polyverts)]
    ctx.clear(0)
    ctx.submit_object_dev(db[0], db[1], db[2], b.render_type, b.log_focus, matrix=b.matrix, max_edges=3)
    ctx.clear(0x0a0a0a)
    ctx.submit_object_dev(da[0], da[1], da[2], a.render_type, a.log_focus, matrix=a.matrix, max_edges=3)
    fc = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    fz = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    ctx.bind_framebuffer(fc, fz)
    ctx.clear(0x050505)
    ctx.submit_object_dev(db[0], db[1], db[2], b.render_type, b.log_focus, matrix=b.matrix, max_edges=3)
    ctx.flush()
    orc = _oracle_frame(P, W, H, 0x050505, [b])
    assert (fz.cpu().numpy() == orc.zbuf()).all() and (fc.cpu().numpy() == orc.cbuf()).all()
    ctx.bind_framebuffer(None, None)
    c, z = ctx.readback()
    orc = _oracle_frame(P, W, H, 0x0a0a0a, [a])
    assert (z == orc.zbuf()).all() and (c == orc.cbuf()).all()
    ctx.submit_object_dev(db[0], db[1], db[2], b.render_type, b.log_focus, matrix=b.matrix, max_edges=3)
    c, z = ctx.readback()
    orc = _oracle_frame(P, W, H, 0x0a0a0a, [a, b])
    assert (z == orc.zbuf()).all() and (c == orc.cbuf()).all()


def test_wire_objects_over_polygons_across_frames(env):
    """three frames back to back, each with filled polygons and a wireframe object over them (the
    per-pixel winner words of the lines are shared by the frames in flight): same frames with
    one frame in flight, and the last equals the CPU restatement"""
    import torch
    P, G = env
    W, H = 640, 480
    fills = [S.gen_c2(5000, W, H, 9, radius=14, seed=3400 + f) for f in range(3)]
    wires = [S.gen_fuzz_polys(1500, W, H, 9, 3500 + f, render_type=S.M_FLAT) for f in range(3)]
    got = []
    for overlap in (True, False):
        ctx = G.Context(W, H)
        ctx.set_frame_overlap(overlap)
        keep = []
        for f in range(3):
            sc, wr = fills[f], wires[f]
            d = [torch.from_numpy(a).cuda() for a in (sc.verts, sc.polys, sc.polyverts)]
            dw = [torch.from_numpy(a).cuda() for a in (wr.verts, wr.polys, wr.polyverts, wr.polyaux)]
            keep += [d, dw]
            ctx.clear(0x020202 * f)
            ctx.submit_object_dev(d[0], d[1], d[2], sc.render_type, sc.log_focus, matrix=sc.matrix, max_edges=3)
            ctx.submit_object_dev(dw[0], dw[1], dw[2], S.M_WIRE, wr.log_focus, d_polyaux=dw[3], matrix=wr.matrix)
        got.append(ctx.readback())
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])
    orc = P.Oracle(W, H)
    orc.clear(0x020202 * 2, 255, 255, 255)
    sc, wr = fills[2], wires[2]
    orc.render_polys(P.transform(sc.matrix, sc.verts), sc.polys, None, sc.polyverts, sc.render_type, sc.log_focus)
    orc.render_polys(P.transform(wr.matrix, wr.verts), wr.polys, wr.polyaux, wr.polyverts, S.M_WIRE, wr.log_focus)
    # unclipped lines over polygons are exact; a clipped end point's depth is a call-history leftover in the
    # reference, so the comparison with the restatement is on the depth buffer (lines never write it) and on
    # the two GPU runs above for colour
    assert np.array_equal(got[0][1], orc.zbuf())


@pytest.mark.parametrize("overlap", [True, False])
def test_host_line_lists_across_frames_without_a_flush(env, overlap):
    """clear, polygons, host lines, clear, other host lines, ... with no flush in between: every
    list is staged in one device array, which a later list must not overwrite while the line
    kernels of the frame before still read it (they run on the other fill stream).  The depth
    tests counted over all frames say whether each frame drew its own lines."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_golden as MG
    P, G = env
    W, H = MG.WIRE_W, MG.WIRE_H
    sc = S.gen_c2(3000, W, H, 7, radius=20, seed=515)
    lists = [MG.wire_lines(600 + f, 2500 + 300 * f, 1, -2, -6200, -100) for f in range(6)]
    # expected depth tests: polygons' fragments + one test per point of each unclipped line
    orc = P.Oracle(W, H)
    want = 0
    for f, lines in enumerate(lists):
        orc.clear(0, 0, 0, 0)
        orc.reset_counters()
        orc.render_polys(P.transform(sc.matrix, sc.verts), sc.polys, None, sc.polyverts, sc.render_type, sc.log_focus)
        for l in lines:
            orc.g_line(l[0:3], l[3:6], int(l[6]), 255, 255, 255)
        want += orc.counters()["fragments"]
    ctx = G.Context(W, H)
    ctx.set_frame_overlap(overlap)
    ctx.reset_counters()
    for f, lines in enumerate(lists):
        ctx.clear(0)
        ctx.submit_scene(sc)
        l = lines.copy()
        l[:, 6] = [P.cl_light(int(c), 255, 255, 255) for c in l[:, 6]]
        ctx.submit_lines(l)
    c, z = ctx.readback()
    assert (z == orc.zbuf()).all() and (c == orc.cbuf()).all()
    assert ctx.counters()["fragments"] == want
    ctx.close()
