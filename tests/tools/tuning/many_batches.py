import sys, time, os
sys.path.insert(0, os.getcwd())
os.environ["GPU_MAX_HW_QUEUES"] = "8"
import torch
from raytracing_amd import gpu as G, scenes as S
sc = S.gen_c2(1_000_000, 1920, 1080, 10)
dev = torch.device("cuda", 0)
dv, dp, dpv = (torch.from_numpy(x).to(dev) for x in (sc.verts, sc.polys, sc.polyverts))
ctx = G.Context(sc.width, sc.height, device=0)
for nb in (1, 10, 100, 1000):
    per = 1_000_000 // nb
    parts = [dp[i * per:(i + 1) * per] for i in range(nb)]
    def frame():
        ctx.clear(0)
        for p in parts:
            ctx.submit_object_dev(dv, p, dpv, sc.render_type, sc.log_focus, matrix=sc.matrix, max_edges=3)
    for _ in range(3): frame()
    ctx.flush(); torch.cuda.synchronize()
    n = 20 if nb < 1000 else 5
    t0 = time.perf_counter()
    for _ in range(n): frame()
    ctx.flush(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%4d batches of %7d triangles a frame: %.3f ms per frame, %.1f us per batch, %.0f Mtris/s" % (nb, per, dt * 1e3, dt * 1e6 / nb, 1.0 / dt), flush=True)
c, z = ctx.readback()
print("fragments", ctx.counters()["fragments"])
