"""draw-order fill: frames of the first k polygons of scenes of wide polygons, for many k, against the CPU
restatement -- what a polygon writes is checked before later polygons cover it"""
import os, sys, time, numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before HIP starts: two frames in flight by default
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import oracle.port as P
from raytracing_amd import scenes as S, gpu as G
bad = total = 0
t0 = time.time()
W, H = 640, 96
for seed, radius, lin in ((1, 150, 100), (2, 400, 100), (3, 60, 70), (4, 250, 70), (5, 30, 100), (6, 500, 90)):
    sc = S.gen_c3(400, W, H, 8, radius=radius, equal_lights=False, lin_percent=lin, seed=0xC0FFEE00 + seed, regular_mapping=True)
    ctx = G.Context(W, H)
    orc = P.Oracle(W, H)
    for h, (tex, lg) in enumerate(sc.textures):
        ctx.texture_upload(h, tex, lg); orc.set_texture(h, tex, lg)
    ctx.set_textured_order(1 | 4)
    vv = P.transform(sc.matrix, sc.verts)
    for k in list(range(1, 130)) + list(range(130, 400, 9)):
        orc.clear(0, 0, 0, 0)
        orc.render_polys(vv, sc.polys[:k], sc.polyaux[:k], sc.polyverts, sc.render_type, sc.log_focus, sc.force_linear)
        ctx.clear(0)
        ctx.submit_object(sc.verts, sc.polys[:k], sc.polyverts, sc.render_type, sc.log_focus, polyaux=sc.polyaux[:k], matrix=sc.matrix,
                          force_linear=sc.force_linear, max_edges=3)
        c, z = ctx.readback(check=False)
        total += 1
        if not (np.array_equal(z, orc.zbuf()) and np.array_equal(c, orc.cbuf())):
            bad += 1
            print("MISMATCH seed", seed, "radius", radius, "k", k, "z px", int((z != orc.zbuf()).sum()), "c px", int((c != orc.cbuf()).sum()), flush=True)
    ctx.close()
    print("scene", seed, "done, bad so far", bad, f"{time.time()-t0:.0f} s", flush=True)
print("TOTAL", total, "bad", bad)
