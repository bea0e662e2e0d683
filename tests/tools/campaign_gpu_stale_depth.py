"""Campaign, not collected by pytest: `python tests/tools/campaign_gpu_stale_depth.py N` draws N random sequences of
filled objects, screen-level polygon lists, wireframe objects and line lists that the screen clips -- the frames in
which a clipped line's depth is a leftover of the polygon clips before it (Clipp2D.cpp:99-104) -- with the reference
build (every case in a child process of its own, started before this process touches the GPU: some fuzz inputs stop
the reference with an arithmetic fault) and through the host mirror with HW_track_clipper_stores(1), whole objects
and polygon by polygon in turn, and compares the frames.  Every case first overwrites all twelve clipper stores, as
the golden cases do.  Needs oracle/_ref (built by __graft_entry__.build()).
Last run: see the line this prints at its end."""
import ctypes as C
import os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from raytracing_amd import scenes as S
import make_golden as MG
import oracle.ref as R

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
W, H = MG.WIRE_W, MG.WIRE_H


def make_case(it):
    rng = np.random.default_rng(31000 + it)
    t = np.array([[-W - 3 - it % 7, -H - 3, -11], [2 * W + 5, -H - 9, -23 - it % 5], [2 * W + 9, 2 * H + 1, -37],
                  [-W - 1, 2 * H + 11, -5], [0, 0, 0]], dtype=np.int32)
    t[4] = t[0]
    steps = [("clear", int(rng.integers(0, 0xFFFFFF))), ("gpolys", S.M_FLAT, [(4, t, int(rng.integers(0, 0xFFFFFF)))])]
    tex_scene = None
    for b in range(int(rng.integers(3, 8))):
        kind = int(rng.integers(0, 7))
        seed = 600000 + it * 16 + b
        obj = tuple(int(v) for v in (rng.integers(0, 256), rng.integers(0, 256), rng.integers(0, 256), rng.integers(-6, 7),
                                     rng.integers(-6, 7), rng.integers(10, 60)))
        me = int(rng.integers(3, 7))
        n = int(rng.integers(20, 1000 // me))          # fewer than M_MAX_OBJECT_VERTICES vertices an object
        if kind == 0:
            steps.append(("object", S.gen_fuzz_polys(n, W, H, 7, seed, render_type=S.M_FLAT, max_edges=me), S.M_FLAT, obj))
        elif kind == 1:
            steps.append(("object", S.gen_fuzz_polys(n, W, H, 7, seed, render_type=S.M_SHADED, max_edges=me), S.M_SHADED, obj))
        elif kind == 2:
            if tex_scene is None:       # one texture per case (handle 0), as in the golden cases
                tex_scene = S.gen_fuzz_polys(n, W, H, 7, seed, render_type=S.M_TEXTURED, max_edges=me, textured=True, equal_lights=True,
                                             zrange=(12, 2000))
            steps.append(("object", tex_scene, S.M_TEXTURED, obj))
        elif kind == 3:
            steps.append(("object", S.gen_fuzz_polys(n, W, H, 7, seed, render_type=S.M_SHADED, max_edges=me), S.M_WIRE, obj))
        elif kind == 4:
            dim = (3, 6)[int(rng.integers(0, 2))]
            steps.append(("gpolys", {3: S.M_FLAT, 6: S.M_SHADED}[dim], S.gen_screen_polys(int(rng.integers(5, 80)), W, H, seed=seed, dim=dim)))
        else:
            steps.append(("lines", MG.wire_lines(seed, int(rng.integers(20, 300)), -int(rng.integers(10, 250)), int(rng.integers(10, 250)),
                                                 -int(rng.integers(1000, 50000)), int(rng.integers(-500, 200)))))
    return steps


def reference_frames(steps, path):
    """in a child process: the frames, or nothing if the reference stopped"""
    pid = os.fork()
    if pid == 0:
        try:
            r = R.Ref(W, H)
            for hnd, (tex, lg) in enumerate(MG._textures_of(steps)):
                r.set_texture(hnd, tex, lg)
            c, z = MG.run_wire_case(steps, r)
            np.savez(path, c=c, z=z)
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
cases, refs = [], []
tmp = tempfile.mkdtemp()
for it in range(N):
    steps = make_case(it)
    cases.append(steps)
    refs.append(reference_frames(steps, os.path.join(tmp, f"{it}.npz")))
    if it % 50 == 49:
        print("reference frames", it + 1, f"{time.time() - t0:.0f} s", flush=True)
print("reference stopped on", sum(r is None for r in refs), "of", N, "cases", flush=True)

import torch  # noqa: E402,F401  (the GPU from here on)
import test_gpu_stale_depth as T  # noqa: E402
from raytracing_amd import _lib  # noqa: E402
_lib.lib()
L = C.CDLL(os.path.join(os.path.dirname(_lib.SO_PATH), "liblt_engine.so"))
L.lte_cbuf.restype = C.POINTER(C.c_int32)
L.lte_zbuf.restype = C.POINTER(C.c_int32)
L.lte_init(W, H)
L.lte_track_clipper_stores(1)
bad = total = 0
for it, (steps, ref) in enumerate(zip(cases, refs)):
    if ref is None:
        continue
    for whole in (it % 2 == 0, it % 2 == 1):
        c, z = T._run(L, steps, whole)
        total += 1
        if not (np.array_equal(c, ref[0]) and np.array_equal(z, ref[1])):
            bad += 1
            print("MISMATCH case", it, "whole objects" if whole else "polygon by polygon", [s[0] + (":%d" % s[2] if s[0] == "object" else "") for s in steps],
                  "c px", int((c != ref[0]).sum()), "z px", int((z != ref[1]).sum()), flush=True)
    if it % 50 == 49:
        print("compared", it + 1, "bad", bad, f"{time.time() - t0:.0f} s", flush=True)
L.lte_track_clipper_stores(0)
print("TOTAL", total, "frames against the reference build's,", bad, "differing;", sum(r is None for r in refs), "cases skipped (reference stopped)")
