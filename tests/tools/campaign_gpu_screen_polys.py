"""Campaign, not collected by pytest: `python tests/tools/campaign_gpu_screen_polys.py N` draws N lists of already-
projected polygons through lt_submit_screen_polys / lt_submit_screen_polys_prp and compares with the CPU restatement.
Last run: 1200 cases, 0 differing."""
import os, sys, time, numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before HIP starts: two frames in flight by default
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import oracle.port as P
from raytracing_amd import scenes as S, gpu as G
KIND = {3: 2, 6: 4, 8: 8}
bad = total = 0
t0 = time.time()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for it in range(N):
    W, H = [(200, 150), (97, 61), (320, 64), (64, 300)][it % 4]
    dim = (3, 6, 8, 8)[it % 4]
    prp = it % 4 == 3
    seed = 900000 + it
    polys = S.gen_screen_polys(40 + it % 90, W, H, seed=seed, dim=dim, max_edges=3 + it % 6, spread=1 + it % 3)
    orc = P.Oracle(W, H)
    ctx = G.Context(W, H)
    orc.clear(0x334455, 200, 200, 200)
    ctx.clear(P.cl_light(0x334455, 200, 200, 200))
    if dim == 8:
        tex = S.make_texture(9 + it % 3, 6)
        orc.set_texture(0, tex, 6)
        ctx.texture_upload(0, tex, 6)
        if it % 2 == 0:
            for (e, t, col) in polys:
                t[:, 3:6] = t[0, 3:6]
        ctx.set_textured_order((0, 1)[it % 2])
    if not prp:
        for (e, t, col) in polys:
            orc.g_polygon(KIND[dim], t, colour=col, r=200 if dim == 3 else 0, g=180 if dim == 3 else 0, b=90 if dim == 3 else 0)
        if dim == 3:
            pl = [(e, t, P.cl_light(col, 200, 180, 90)) for (e, t, col) in polys]
        elif dim == 8:
            pl = [(e, t, col, 0) for (e, t, col) in polys]
        else:
            pl = polys
        ctx.submit_screen_polys(KIND[dim], pl)
    else:
        rng = S.Rng(seed + 7)
        pl = []
        for (e, t, col) in polys:
            O = np.zeros(8, np.int32)
            O[0:2] = rng.ints(-400, 400, 2); O[2] = rng.ints(20, 1500, 1)[0]; O[6:8] = rng.ints(0, 1023, 2) << 16
            mag = (3, 40, 300)[it % 3]
            u = rng.ints(-mag, mag, 3).astype(np.int32); v = rng.ints(-mag, mag, 3).astype(np.int32)
            ls = int(rng.ints(4, 9, 1)[0])
            orc.g_prp_polygon(t, O, u, v, 0, ls, 7)
            pl.append((e, t, O, u, v, 0, ls))
        ctx.submit_screen_polys_prp(pl, 7)
    try:
        c, z = ctx.readback(check=False)
    except G.LtError as ex:
        print("error", it, ex); c, z = ctx.readback(check=False)
    ok = np.array_equal(z, orc.zbuf()) and np.array_equal(c, orc.cbuf())
    total += 1
    if not ok:
        bad += 1
        print("MISMATCH it", it, "dim", dim, "prp", prp, W, H, "z px", int((z != orc.zbuf()).sum()), "c px", int((c != orc.cbuf()).sum()), flush=True)
    ctx.close()
    if it % 100 == 99:
        print("done", it + 1, "bad", bad, f"{time.time() - t0:.0f} s", flush=True)
print("TOTAL", total, "bad", bad)
