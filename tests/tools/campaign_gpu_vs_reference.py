"""Campaign, not collected by pytest: `python tests/tools/campaign_gpu_vs_reference.py N` draws N frames with the
reference build (oracle/_ref, every frame in a child process started before this process touches the GPU) and with the
library through the C-ABI (device arrays), and compares colour and depth pixel for pixel: triangle soups of the
benchmark's kinds (Gouraud, textured with equal and with differing vertex lights -- the latter also with enough
triangles for the per-band lists of the draw-order fill --, a tenth perspective-mapped) at 640x480 and 1920x1080,
radii 2..200, under the animated batch's per-frame matrices (triangles through the near plane and the screen borders),
two objects per frame; fuzz polygon objects at 160x120 and 200x150.  Needs oracle/_ref (built by
__graft_entry__.build()).  Last run (400): 389 frames, 0 differing; the reference stopped on 11."""
import os, sys, time, tempfile
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before HIP starts: two frames in flight by default
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from raytracing_amd import scenes as S, frames as F
import oracle.ref as R
import oracle.port as P

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40


def make_case(it):
    rng = np.random.default_rng(52000 + it)
    objs = []
    if it % 4 == 3:
        W, H, lf = ((160, 120, 7), (200, 150, 7))[(it // 4) % 2]
        for b in range(2):
            rt = (S.M_FLAT, S.M_SHADED, S.M_TEXTURED)[int(rng.integers(0, 3))]
            sc = S.gen_fuzz_polys(int(rng.integers(100, 3000)), W, H, lf, 730000 + it * 4 + b, render_type=rt, max_edges=int(rng.integers(3, 7)),
                                  textured=rt == S.M_TEXTURED, equal_lights=bool(rng.integers(0, 2)), zrange=(12, int(rng.integers(300, 3000))))
            objs.append((sc, rt, sc.matrix))
        tex = (S.make_texture(9, 6), 6)
    else:
        W, H, lf = ((640, 480, 9), (1920, 1080, 10))[it % 2]
        for b in range(2):
            kind = int(rng.integers(0, 4))
            radius = int((2, 4, 8, 16, 40, 200)[int(rng.integers(0, 6))])
            n = int(rng.integers(2000, 60000 if radius <= 16 else 4000))
            if kind == 3:
                n = int(rng.integers(21000, 50000)); radius = min(radius, 16)
            seed = 740000 + it * 4 + b
            if kind == 0:
                sc = S.gen_c2(n, W, H, lf, radius=radius, seed=seed)
            else:
                sc = S.gen_c3(n, W, H, lf, radius=radius, seed=seed, equal_lights=kind == 1, lin_percent=90, regular_mapping=True)
            m = F.frame_matrix(int(rng.integers(0, 1024))) if rng.integers(0, 3) else sc.matrix
            objs.append((sc, sc.render_type, m))
        tex = (S.make_texture(11, 8), 8)
    return W, H, tex, objs


def reference_frame(W, H, tex, objs, path):
    pid = os.fork()
    if pid == 0:
        try:
            r = R.Ref(W, H)
            r.set_texture(0, tex[0], tex[1])
            r.clear(0x102030, 255, 255, 255)
            for (sc, rt, m) in objs:
                r.render_polys(P.transform(m, sc.verts), sc.polys, sc.polyaux, sc.polyverts, rt, sc.log_focus)
            np.savez(path, c=r.cbuf(), z=r.zbuf())
        finally:
            os._exit(0)
    _, status = os.waitpid(pid, 0)
    if status != 0 or not os.path.exists(path):
        return None
    d = np.load(path)
    out = (d["c"].copy(), d["z"].copy())
    os.remove(path)
    return out


t0 = time.time()
tmp = tempfile.mkdtemp()
cases, refs = [], []
for it in range(N):
    W, H, tex, objs = make_case(it)
    cases.append((W, H, tex, objs))
    refs.append(reference_frame(W, H, tex, objs, os.path.join(tmp, f"{it}.npz")))
    print("reference frame", it, W, H, [(sc.name, rt) for (sc, rt, m) in objs], "stopped" if refs[-1] is None else "", f"{time.time() - t0:.0f} s", flush=True)

import torch  # noqa: E402  (the GPU from here on)
from raytracing_amd import gpu as G  # noqa: E402
bad = total = 0
for it, ((W, H, tex, objs), ref) in enumerate(zip(cases, refs)):
    if ref is None:
        continue
    ctx = G.Context(W, H)
    ctx.texture_upload(0, tex[0], tex[1])
    if it % 5 == 1: ctx.set_tile_width(128)
    if it % 5 == 2: ctx.set_tile_width(64)
    if it % 7 == 3: ctx.set_pipelining(False)
    ctx.clear(P.cl_light(0x102030, 255, 255, 255))
    keep = []
    for (sc, rt, m) in objs:
        d = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (sc.verts, sc.polys, sc.polyverts)]
        pa = torch.from_numpy(np.ascontiguousarray(sc.polyaux)).cuda() if sc.polyaux is not None else None
        keep.append((d, pa))
        ctx.submit_object_dev(d[0], d[1], d[2], rt, sc.log_focus, d_polyaux=pa, matrix=m, max_edges=min(int(sc.polys[:, 2].max()), 15))
    c, z = ctx.readback(check=False)
    ctx.close()
    total += 1
    if not (np.array_equal(c, ref[0]) and np.array_equal(z, ref[1])):
        bad += 1
        print("MISMATCH frame", it, W, H, [(sc.name, rt) for (sc, rt, m) in objs], "c px", int((c != ref[0]).sum()), "z px", int((z != ref[1]).sum()), flush=True)
print("TOTAL", total, "frames against the reference build's,", bad, "differing;", sum(r is None for r in refs), "skipped (reference stopped)")
