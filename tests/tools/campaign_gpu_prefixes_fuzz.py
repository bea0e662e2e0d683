"""Campaign, not collected by pytest: frames of the first k polygons (k = 1..100) of N fuzz scenes of every render type against\nthe CPU restatement: a draw is checked before later polygons cover it.  Last run: N = 120, 12000 frames, 0 differing."""
import os, sys, time, numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before HIP starts: two frames in flight by default
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import oracle.port as P
from raytracing_amd import scenes as S, gpu as G
bad = total = 0
t0 = time.time()
sizes = [(160, 120, 7), (320, 200, 8), (97, 61, 6), (256, 256, 8)]
for it in range(int(sys.argv[1])):
    rng = np.random.default_rng(9000 + it)
    W, H, lf = sizes[it % len(sizes)]
    rt = (S.M_FLAT, S.M_SHADED, S.M_TEXTURED, S.M_TEXTURED)[it % 4]
    tex = rt == S.M_TEXTURED
    n = 100
    sc = S.gen_fuzz_polys(n, W, H, lf, 910000 + it, render_type=rt, max_edges=int(rng.integers(3, 9)), textured=tex,
                          equal_lights=(it % 8 < 4), zrange=(12, int(rng.integers(200, 3000))))
    ctx = G.Context(W, H)
    orc = P.Oracle(W, H)
    for h, (t_, lg) in enumerate(sc.textures):
        ctx.texture_upload(h, t_, lg); orc.set_texture(h, t_, lg)
    if tex:
        ctx.set_textured_order((0, 1 | 4, 1 | 8, 0)[(it // 4) % 4])
    ctx.set_tile_width((0, 64, 128)[it % 3])
    vv = P.transform(sc.matrix, sc.verts)
    me = int(sc.polys[:, 2].max())
    for k in range(1, n + 1):
        orc.clear(0x102030, 255, 255, 255)
        orc.render_polys(vv, sc.polys[:k], None if sc.polyaux is None else sc.polyaux[:k], sc.polyverts, rt, sc.log_focus, sc.force_linear)
        ctx.clear(P.cl_light(0x102030, 255, 255, 255))
        ctx.submit_object(sc.verts, sc.polys[:k], sc.polyverts, rt, sc.log_focus, polyaux=None if sc.polyaux is None else sc.polyaux[:k],
                          matrix=sc.matrix, force_linear=sc.force_linear, max_edges=min(me, 15))
        c, z = ctx.readback(check=False)
        total += 1
        if not (np.array_equal(z, orc.zbuf()) and np.array_equal(c, orc.cbuf())):
            bad += 1
            print("MISMATCH it", it, "rt", rt, W, H, "k", k, "z px", int((z != orc.zbuf()).sum()), "c px", int((c != orc.cbuf()).sum()), flush=True)
            break
    ctx.close()
    if it % 10 == 9: print("done", it + 1, "bad", bad, f"{time.time()-t0:.0f} s", flush=True)
print("TOTAL", total, "bad", bad)
