"""Headline loop (clear + one object of 1M triangles, 200 frames) with the caller's stream a non-blocking one; see README.md."""
import os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("LT_Q", "8")
sys.path.insert(0, os.getcwd())
import torch
from raytracing_amd import gpu as G, scenes as S
mode = os.environ.get("LT_MODE", "plain")
sc = S.gen_c2(1_000_000, 1920, 1080, 10, radius=8)
dev = torch.device("cuda", 0)
dv = torch.from_numpy(sc.verts).to(dev); dp = torch.from_numpy(sc.polys).to(dev); dpv = torch.from_numpy(sc.polyverts).to(dev)
da = torch.from_numpy(sc.polyaux).to(dev) if sc.polyaux is not None else None
ctx = G.Context(sc.width, sc.height, device=0)
side = torch.cuda.Stream()
if os.environ.get("LT_SIDE_STREAM", "1") == "1":
    ctx.set_stream(side.cuda_stream)
def step():
    ctx.clear(0)
    ctx.submit_object_dev(dv, dp, dpv, sc.render_type, sc.log_focus, d_polyaux=da, matrix=sc.matrix, force_linear=sc.force_linear, max_edges=3)
if mode == "probe":
    ctx.set_profiling(True)
    ctx.set_pipelining(False)
    for i in range(4):
        step()
        if i == 0: ctx.stage_ms()
    ctx.stage_ms()
    ctx.set_pipelining(True)
    for _ in range(5): step()
    ctx.reset_counters()
    ctx.set_profiling("fill", every=4)
    ctx.stage_ms()
elif mode == "noflush":
    for _ in range(8): step()
else:
    for _ in range(8): step()
    ctx.flush()
torch.cuda.synchronize()
K = 200
t0 = time.perf_counter()
for _ in range(K): step()
torch.cuda.synchronize()
print(sys.argv[1] if len(sys.argv) > 1 else "", mode, round((time.perf_counter() - t0) / K * 1e3, 4), flush=True)
