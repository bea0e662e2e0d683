"""Campaign, not collected by pytest: `python tests/tools/campaign_repeat_mixed.py N REPS` builds N frames out of several
batches each (fuzz objects of every render type, textured ones with differing vertex lights in both draw-order forms,
screen-level polygon lists, line lists, wireframe objects; host-array and device-array submits; both tile widths; with
and without pipelining and frame overlap), checks the first render of each against the CPU restatement and then renders
the frame REPS times more, comparing every result with the first.  A rare run-to-run difference (a race between
batches, streams or work sets) shows up as a count.  Last run (60 300): 60 frames, all equal to the CPU restatement; 18000 repeat renders, 0 differing."""
import os, sys, time, numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before HIP starts: two frames in flight by default
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import oracle.port as P
from raytracing_amd import scenes as S, gpu as G

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 100
SIZES = [(320, 200, 8), (640, 480, 9), (1000, 700, 9), (160, 120, 7), (1920, 90, 10)]
KINDS = ("flat", "shaded", "textured-equal", "textured-q2", "textured-q2-lists", "screen", "lines", "wire")


def make_frame(it):
    rng = np.random.default_rng(77000 + it)
    W, H, lf = SIZES[it % len(SIZES)]
    steps = []
    for b in range(int(rng.integers(2, 6))):
        kind = KINDS[int(rng.integers(0, len(KINDS)))]
        seed = 900000 + it * 16 + b
        me = int(rng.integers(3, 8))
        zr = (12, int(rng.integers(300, 2500)))
        if kind == "flat":
            steps.append(("object", S.gen_fuzz_polys(int(rng.integers(50, 4000)), W, H, lf, seed, render_type=S.M_FLAT, max_edges=me, zrange=zr), S.M_FLAT))
        elif kind == "shaded":
            steps.append(("object", S.gen_fuzz_polys(int(rng.integers(50, 4000)), W, H, lf, seed, render_type=S.M_SHADED, max_edges=me, zrange=zr), S.M_SHADED))
        elif kind.startswith("textured"):
            n = int(rng.integers(20000, 26000)) if kind.endswith("lists") else int(rng.integers(50, 3000))
            steps.append(("object", S.gen_fuzz_polys(n, W, H, lf, seed, render_type=S.M_TEXTURED, max_edges=3 if n > 10000 else me, textured=True,
                                                     equal_lights=kind.endswith("equal"), zrange=zr), S.M_TEXTURED))
        elif kind == "wire":
            steps.append(("object", S.gen_fuzz_polys(int(rng.integers(20, 400)), W, H, lf, seed, render_type=S.M_SHADED, max_edges=me, zrange=zr), S.M_WIRE))
        elif kind == "screen":
            dim = (3, 6, 8)[int(rng.integers(0, 3))]
            steps.append(("gpolys", {3: S.M_FLAT, 6: S.M_SHADED, 8: S.M_TEXTURED}[dim], S.gen_screen_polys(int(rng.integers(10, 300)), W, H, seed=seed, dim=dim)))
        else:
            n = int(rng.integers(10, 400))
            l = np.zeros((n, 8), np.int32)
            l[:, 0] = rng.integers(-W // 2, W + W // 2, n); l[:, 3] = rng.integers(-W // 2, W + W // 2, n)
            l[:, 1] = rng.integers(-H // 2, H + H // 2, n); l[:, 4] = rng.integers(-H // 2, H + H // 2, n)
            l[:, 2] = rng.integers(-50000, 0, n); l[:, 5] = rng.integers(-50000, 0, n)
            l[:, 6] = rng.integers(0, 0xFFFFFF, n)
            steps.append(("lines", l))
    return W, H, lf, steps


def draw_oracle(W, H, steps, tex):
    orc = P.Oracle(W, H)
    orc.set_texture(0, tex, 6)
    orc.clear(0x203040, 255, 255, 255)
    for st in steps:
        if st[0] == "object":
            sc = st[1]
            orc.render_polys(P.transform(sc.matrix, sc.verts), sc.polys, sc.polyaux, sc.polyverts, st[2], sc.log_focus)
        elif st[0] == "gpolys":
            for (e, t, col) in st[2]:
                orc.g_polygon(st[1], t, colour=col, r=255, g=255, b=255)
        else:
            for l in st[1]:
                orc.g_line(l[0:3], l[3:6], int(l[6]), 255, 255, 255)
    return orc.cbuf().copy(), orc.zbuf().copy()


def draw_gpu(ctx, steps, dev):
    ctx.clear(P.cl_light(0x203040, 255, 255, 255))
    for k, st in enumerate(steps):
        if st[0] == "object":
            sc = st[1]
            me = min(int(sc.polys[:, 2].max()), 15)
            if dev[k] is not None:
                v, po, pv, pa = dev[k]
                ctx.submit_object_dev(v, po, pv, st[2], sc.log_focus, d_polyaux=pa, matrix=sc.matrix, max_edges=me)
            else:
                ctx.submit_object(sc.verts, sc.polys, sc.polyverts, st[2], sc.log_focus, polyaux=sc.polyaux, matrix=sc.matrix, max_edges=me)
        elif st[0] == "gpolys":
            polys = st[2]
            if st[1] == S.M_FLAT:
                polys = [(e, t, P.cl_light(col, 255, 255, 255)) for (e, t, col) in polys]
            ctx.submit_screen_polys(st[1], polys)
        else:
            ctx.submit_lines(st[1])
    return ctx.readback(check=False)


tex = S.make_texture(9, 6)
bad_first = bad_repeat = frames = 0
t0 = time.time()
for it in range(N):
    W, H, lf, steps = make_frame(it)
    oc, oz = draw_oracle(W, H, steps, tex)
    ctx = G.Context(W, H)
    ctx.texture_upload(0, tex, 6)
    if it % 4 == 1: ctx.set_tile_width(128)
    if it % 4 == 2: ctx.set_tile_width(64)
    if it % 7 == 3: ctx.set_pipelining(False)
    if it % 5 == 4: ctx.set_frame_overlap(False)
    dev = []
    for k, st in enumerate(steps):
        if st[0] == "object" and (it + k) % 2 == 0:
            sc = st[1]
            dev.append(tuple(torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (sc.verts, sc.polys, sc.polyverts, sc.polyaux)))
        else:
            dev.append(None)
    c0, z0 = draw_gpu(ctx, steps, dev)
    if not (np.array_equal(c0, oc) and np.array_equal(z0, oz)):
        bad_first += 1
        print("MISMATCH with the CPU restatement: frame", it, W, H, [s[0] + (":" + str(s[2]) if s[0] == "object" else "") for s in steps],
              "c px", int((c0 != oc).sum()), "z px", int((z0 != oz).sum()), flush=True)
    for rep in range(REPS):
        c, z = draw_gpu(ctx, steps, dev)
        frames += 1
        if not (np.array_equal(c, c0) and np.array_equal(z, z0)):
            bad_repeat += 1
            print("  frame", it, "rep", rep, "differs from its first render: c px", int((c != c0).sum()), "z px", int((z != z0).sum()),
                  [s[0] for s in steps], flush=True)
    ctx.close()
    print("frame", it, W, H, [s[0] + (":%d:%d" % (s[2], s[1].polys.shape[0]) if s[0] == "object" else "") for s in steps],
          "bad so far", bad_first, bad_repeat, f"{time.time() - t0:.0f} s", flush=True)
print("TOTAL frames", N, "differing from the CPU restatement", bad_first, "| repeat renders", frames, "differing from the first", bad_repeat)
