"""Campaign, not collected by pytest: prefixes of dense scenes (thousands of polygons a tile: coarse depth table, both tile\nwidths, draw-order lists) against the CPU restatement.  Last run: 316 frames, 0 differing."""
import os, sys, time, numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before HIP starts: two frames in flight by default
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import oracle.port as P
from raytracing_amd import scenes as S, gpu as G
bad = total = 0
t0 = time.time()
W, H = 256, 128
for name, sc in (("shaded", S.gen_c2(40000, W, H, 7, radius=14, seed=0xC0FFEEB1)),
                 ("textured", S.gen_c3(40000, W, H, 7, radius=14, equal_lights=True, lin_percent=90, seed=0xC0FFEEB2, regular_mapping=True)),
                 ("textured q2", S.gen_c3(40000, W, H, 7, radius=14, equal_lights=False, lin_percent=90, seed=0xC0FFEEB3, regular_mapping=True)),
                 ("shaded large", S.gen_c2(6000, W, H, 7, radius=90, seed=0xC0FFEEB4))):
    ctx = G.Context(W, H)
    orc = P.Oracle(W, H)
    for h, (t_, lg) in enumerate(sc.textures):
        ctx.texture_upload(h, t_, lg); orc.set_texture(h, t_, lg)
    vv = P.transform(sc.matrix, sc.verts)
    n = sc.n_polys
    for k in list(range(900, n, max(n // 40, 1))) + [n]:
        orc.clear(0, 0, 0, 0)
        orc.render_polys(vv, sc.polys[:k], None if sc.polyaux is None else sc.polyaux[:k], sc.polyverts, sc.render_type, sc.log_focus, sc.force_linear)
        for tile in (64, 128):
            ctx.set_tile_width(tile)
            ctx.clear(0)
            ctx.submit_object(sc.verts, sc.polys[:k], sc.polyverts, sc.render_type, sc.log_focus, polyaux=None if sc.polyaux is None else sc.polyaux[:k],
                              matrix=sc.matrix, force_linear=sc.force_linear, max_edges=3)
            c, z = ctx.readback(check=False)
            total += 1
            if not (np.array_equal(z, orc.zbuf()) and np.array_equal(c, orc.cbuf())):
                bad += 1
                print("MISMATCH", name, "k", k, "tile", tile, "z px", int((z != orc.zbuf()).sum()), "c px", int((c != orc.cbuf()).sum()), flush=True)
    ctx.close()
    print(name, "done, bad so far", bad, f"{time.time()-t0:.0f} s", flush=True)
print("TOTAL", total, "bad", bad)
