import sys, time, os
sys.path.insert(0, os.getcwd())
os.environ["GPU_MAX_HW_QUEUES"] = "8"
import torch
from raytracing_amd import gpu as G, scenes as S
sc = S.gen_c2(1_000_000, 1920, 1080, 10)
dev = torch.device("cuda", 0)
dv, dp, dpv = (torch.from_numpy(x).to(dev) for x in (sc.verts, sc.polys, sc.polyverts))
ctx = G.Context(sc.width, sc.height, device=0)
def frame():
    ctx.clear(0)
    ctx.submit_object_dev(dv, dp, dpv, sc.render_type, sc.log_focus, matrix=sc.matrix, max_edges=3)
    ctx.flush()
for _ in range(5): frame()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): frame()
torch.cuda.synchronize()
print("one frame at a time: %.4f ms per frame" % ((time.perf_counter() - t0) * 10))
