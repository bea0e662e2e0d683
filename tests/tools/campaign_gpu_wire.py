"""Campaign, not collected by pytest: `python tests/tools/campaign_gpu_wire.py N` draws N fuzz objects as wireframes (M_WIRE,\nclip windows included) through the C-ABI and compares with the CPU restatement.  Last run: 800 cases, 0 differing."""
import os, sys, time, numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before HIP starts: two frames in flight by default
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import oracle.port as P
from raytracing_amd import scenes as S, gpu as G
bad = total = 0
t0 = time.time()
sizes = [(160, 120, 7), (320, 200, 8), (97, 61, 6), (640, 33, 8), (70, 400, 7)]
for it in range(int(sys.argv[1])):
    rng = np.random.default_rng(4000 + it)
    W, H, lf = sizes[it % len(sizes)]
    n = int(rng.integers(20, 300)); me = int(rng.integers(3, 10)); zr = (12, int(rng.integers(200, 3000)))
    sc = S.gen_fuzz_polys(n, W, H, lf, 800000 + it, render_type=S.M_SHADED, max_edges=me, zrange=zr)
    orc = P.Oracle(W, H)
    win = None
    if it % 5 == 0:
        x0, y0 = int(rng.integers(0, W // 2)), int(rng.integers(0, H // 2))
        win = (x0, y0, int(rng.integers(x0 + 1, W)), int(rng.integers(y0 + 1, H)))
        orc.set_clip_window(*win)
    orc.clear(0x102030, 255, 255, 255); orc.reset_counters()
    orc.render_polys(P.transform(sc.matrix, sc.verts), sc.polys, sc.polyaux, sc.polyverts, S.M_WIRE, sc.log_focus)
    ctx = G.Context(W, H)
    if win: ctx.set_clip_window(*win)
    if it % 7 == 0: ctx.set_pipelining(False)
    ctx.clear(P.cl_light(0x102030, 255, 255, 255)); ctx.reset_counters()
    ctx.submit_object(sc.verts, sc.polys, sc.polyverts, S.M_WIRE, sc.log_focus, polyaux=sc.polyaux, matrix=sc.matrix, max_edges=min(me, 15))
    c, z = ctx.readback(check=False)
    n_ = ctx.counters()
    ok = np.array_equal(z, orc.zbuf()) and np.array_equal(c, orc.cbuf()) and n_["fragments"] == orc.counters()["fragments"]
    total += 1
    if not ok:
        bad += 1
        print("MISMATCH it", it, W, H, "c px", int((c != orc.cbuf()).sum()), n_["fragments"], orc.counters()["fragments"], flush=True)
    ctx.close()
    if it % 100 == 99: print("done", it + 1, "bad", bad, f"{time.time()-t0:.0f} s", flush=True)
print("TOTAL", total, "bad", bad)
