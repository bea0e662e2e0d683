"""Campaign, not collected by pytest: `python tests/tools/campaign_gpu_shade.py N` shades N random objects (lt_shade_object)\nand compares with the CPU restatement.  Last run: 500 cases, 0 differing."""
import os, sys, time, numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before HIP starts: two frames in flight by default
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import oracle.port as P
from raytracing_amd import scenes as S, gpu as G
bad = total = 0
ctx = G.Context(64, 64)
t0 = time.time()
for it in range(int(sys.argv[1])):
    rng = np.random.default_rng(6000 + it)
    n = int(rng.integers(1, 3000))
    case = S.gen_shading_case(n, 70000 + it, extent=int(rng.choice([200, 2000, 60000])), far_light=bool(it % 3 == 0))
    s9 = P.self_matrix(int(rng.integers(0, 256)), int(rng.integers(0, 256)), int(rng.integers(0, 256)))
    pos = tuple(int(v) for v in rng.integers(-5000, 5000, 3))
    want_pv, want_pa, skipped, nbad = P.shade_polys(s9, pos, case["verts"], case["normals"], case["polys"], case["polyverts"],
                                                    case["polynormals"], None, case["lights"])
    pv, pa = ctx.shade_object(s9, pos, case["verts"], case["normals"], case["polys"], case["polyverts"], case["polynormals"], None, case["lights"])
    try:
        ctx.flush()
    except G.LtError:
        pass
    total += 1
    if not (np.array_equal(pv, want_pv) and np.array_equal(pa[:, :3], want_pa[:, :3])):
        bad += 1
        print("MISMATCH it", it, n, flush=True)
    if it % 100 == 99: print("done", it + 1, "bad", bad, f"{time.time()-t0:.0f} s", flush=True)
print("TOTAL", total, "bad", bad)
