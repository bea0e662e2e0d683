"""Campaign, not collected by pytest: `python tests/tools/campaign_gpu_objects.py N` renders N fuzz objects (all render
types, clip windows, both tile widths, four and eight waves a tile, draw-order modes, with and without pipelining) through the C-ABI and compares
frames and counters with the CPU restatement.  Last run: 5000 cases, 0 differing."""
import os, sys, time, numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before HIP starts: two frames in flight by default
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import oracle.port as P
from raytracing_amd import scenes as S, gpu as G

def check(sc, name, order=None, window=None, tile=0, pipel=True, waves=0):
    W, H = sc.width, sc.height
    orc = P.Oracle(W, H)
    for h, (tex, lg) in enumerate(sc.textures):
        orc.set_texture(h, tex, lg)
    if window:
        orc.set_clip_window(*window)
    orc.clear(0x102030, 255, 255, 255)
    orc.reset_counters()
    orc.render_polys(P.transform(sc.matrix, sc.verts), sc.polys, sc.polyaux, sc.polyverts, sc.render_type, sc.log_focus, sc.force_linear)
    ctx = G.Context(W, H)
    if order is not None:
        ctx.set_textured_order(order)
    if window:
        ctx.set_clip_window(*window)
    if tile:
        ctx.set_tile_width(tile)
    ctx.set_tile_waves(waves)
    if not pipel:
        ctx.set_pipelining(False)
    ctx.clear(P.cl_light(0x102030, 255, 255, 255))
    ctx.reset_counters()
    ctx.submit_scene(sc)
    c, z = ctx.readback(check=False)
    n = ctx.counters()
    ok = np.array_equal(z, orc.zbuf()) and np.array_equal(c, orc.cbuf()) and n["fragments"] == orc.counters()["fragments"] \
        and n["polys_drawn"] == orc.counters()["polys_drawn"]
    ctx.close()
    if not ok:
        print("MISMATCH", name, "z px", int((z != orc.zbuf()).sum()), "c px", int((c != orc.cbuf()).sum()),
              "frags", n["fragments"], orc.counters()["fragments"], "drawn", n["polys_drawn"], orc.counters()["polys_drawn"], flush=True)
    return ok

bad = total = 0
t0 = time.time()
rng = np.random.default_rng(12345)
sizes = [(160, 120, 7), (320, 200, 8), (97, 61, 6), (640, 33, 8), (70, 400, 7), (256, 256, 8)]
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 400):
    W, H, lf = sizes[it % len(sizes)]
    rt = (S.M_FLAT, S.M_SHADED, S.M_TEXTURED)[it % 3]
    tex = rt == S.M_TEXTURED
    seed = 700000 + it
    n = int(rng.integers(20, 400))
    zr = (12, int(rng.integers(200, 3000)))
    me = int(rng.integers(3, 10))
    sc = S.gen_fuzz_polys(n, W, H, lf, seed, render_type=rt, max_edges=me, textured=tex, equal_lights=bool(it % 2), zrange=zr)
    if tex and it % 5 == 0:
        sc.force_linear = 1
    kw = {}
    if it % 7 == 0:
        x0, y0 = int(rng.integers(0, W // 2)), int(rng.integers(0, H // 2))
        kw["window"] = (x0, y0, int(rng.integers(x0 + 1, W)), int(rng.integers(y0 + 1, H)))
    if tex:
        kw["order"] = (0, 1, 1 | 4, 0 | 4)[it % 4]
    kw["tile"] = (0, 64, 128)[it % 3]
    kw["waves"] = (0, 4, 8)[(it // 3) % 3]
    kw["pipel"] = it % 11 != 0
    total += 1
    if not check(sc, f"it {it} seed {seed} {W}x{H} rt {rt} n {n} me {me} {kw}", **kw):
        bad += 1
    if it % 100 == 99:
        print("done", it + 1, "bad", bad, f"{time.time() - t0:.0f} s", flush=True)
print("TOTAL", total, "bad", bad)
