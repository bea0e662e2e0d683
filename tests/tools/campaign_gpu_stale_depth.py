"""Campaign, not collected by pytest: `python tests/tools/campaign_gpu_stale_depth.py N` draws N random sequences of
filled objects (three textures of different sizes among their polygons), screen-level polygon lists of every kind,
perspective-mapped polygon lists, clip windows, wireframe objects and lists of lines, dots and pixels that the screen
clips -- the frames in which a clipped line's depth is a leftover of the polygon clips before it (Clipp2D.cpp:99-104) -- with the reference
build (every case in a child process of its own, started before this process touches the GPU: some fuzz inputs stop
the reference with an arithmetic fault) and through the host mirror with HW_track_clipper_stores(1), whole objects
and polygon by polygon in turn, and compares the frames.  Every case first overwrites all twelve clipper stores, as
the golden cases do.  Needs oracle/_ref (built by __graft_entry__.build()).
Last run (800): 1592 frames against the reference build's, 0 differing; the reference stopped on 4 sequences."""
import ctypes as C
import os, sys, time, tempfile
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before HIP starts: two frames in flight by default
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from raytracing_amd import scenes as S
import make_golden as MG
import oracle.ref as R
import oracle.port as P

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
W, H = MG.WIRE_W, MG.WIRE_H


def make_case(it):
    """-> (textures [(texels, log size)] for handles 0..2, steps)"""
    rng = np.random.default_rng(31000 + it)
    texs = [(np.ascontiguousarray(S.make_texture(500 + 3 * it + h, lg), dtype=np.int32), lg) for h, lg in enumerate((6, 4, 5))]
    t = np.array([[-W - 3 - it % 7, -H - 3, -11], [2 * W + 5, -H - 9, -23 - it % 5], [2 * W + 9, 2 * H + 1, -37],
                  [-W - 1, 2 * H + 11, -5], [0, 0, 0]], dtype=np.int32)
    t[4] = t[0]
    steps = [("clear", int(rng.integers(0, 0xFFFFFF))), ("gpolys", S.M_FLAT, [(4, t, int(rng.integers(0, 0xFFFFFF)), 0)])]
    for b in range(int(rng.integers(3, 9))):
        kind = int(rng.integers(0, 10))
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
            sc = S.gen_fuzz_polys(n, W, H, 7, seed, render_type=S.M_TEXTURED, max_edges=me, textured=True, equal_lights=bool(rng.integers(0, 2)),
                                  zrange=(12, 2000))
            sc.polys[:, 1] = rng.integers(0, 3, sc.polys.shape[0])
            steps.append(("object", sc, S.M_TEXTURED, obj))
        elif kind == 3:
            steps.append(("object", S.gen_fuzz_polys(n, W, H, 7, seed, render_type=S.M_SHADED, max_edges=me), S.M_WIRE, obj))
        elif kind == 4:
            dim = (3, 6, 8)[int(rng.integers(0, 3))]
            steps.append(("gpolys", {3: S.M_FLAT, 6: S.M_SHADED, 8: S.M_TEXTURED}[dim],
                          [(e, tt, col, int(rng.integers(0, 3))) for (e, tt, col) in S.gen_screen_polys(int(rng.integers(5, 80)), W, H, seed=seed, dim=dim)]))
        elif kind == 5:
            polys = []
            for (e, tt, col) in S.gen_screen_polys(int(rng.integers(5, 40)), W, H, seed=seed, dim=8, max_edges=6):
                O = np.zeros(8, np.int32)
                O[0:2] = rng.integers(-400, 400, 2); O[2] = rng.integers(20, 1500); O[6:8] = rng.integers(0, 1023, 2) << 16
                mag = (3, 40, 300)[int(rng.integers(0, 3))]
                u = rng.integers(-mag, mag, 3).astype(np.int32); v = rng.integers(-mag, mag, 3).astype(np.int32)
                polys.append((e, tt, O, u, v, int(rng.integers(0, 3)), int(rng.integers(4, 9))))
            steps.append(("prp", polys))
        elif kind == 6:
            x0, y0 = int(rng.integers(0, W // 2)), int(rng.integers(0, H // 2))
            steps.append(("window", (x0, y0, int(rng.integers(x0 + 1, W)), int(rng.integers(y0 + 1, H)))))
        else:
            l = MG.wire_lines(seed, int(rng.integers(20, 300)), -int(rng.integers(10, 250)), int(rng.integers(10, 250)),
                              -int(rng.integers(1000, 50000)), int(rng.integers(-500, 200)))
            l[:, 7] = rng.choice([0, 0, 0, 0, 1, 2], l.shape[0])          # a few dots and pixels among the lines
            off = (l[:, 0] < 0) | (l[:, 0] >= W) | (l[:, 1] < 0) | (l[:, 1] >= H)
            l[off & (l[:, 7] == 2), 7] = 0                                  # G_pixel does not clip (GraphicsBase.cpp:117-125)
            steps.append(("lines", l))
    return texs, steps


def run_reference(r, texs, steps):
    for h, (tex, lg) in enumerate(texs):
        r.set_texture(h, tex, lg)
    try:
        for st in steps:
            if st[0] == "clear":
                r.clear(st[1], 255, 255, 255)
            elif st[0] == "window":
                r.set_clip_window(*st[1])
            elif st[0] == "gpolys":
                for (e, t, col, tex) in st[2]:
                    r.g_polygon(st[1], t, colour=col, tex=tex)
            elif st[0] == "prp":
                for (e, t, O, u, v, tex, ls) in st[1]:
                    r.g_prp_polygon(t, O, u, v, tex, ls, 7)
            elif st[0] == "object":
                _, sc, rt, obj = st
                m = P.make_matrix((0, 0, 0, 0, 0, 0), obj, 0)
                r.render_polys(P.transform(m, sc.verts), sc.polys, sc.polyaux, sc.polyverts, rt, sc.log_focus)
            else:
                for l in st[1]:
                    if l[7] == 0:
                        r.g_line(l[0:3], l[3:6], int(l[6]), 255, 255, 255)
                    elif l[7] == 1:
                        r.g_dot(l[0:3], int(l[6]), 255, 255, 255)
                    else:
                        r.g_pixel(l[0:3], int(l[6]))
        return r.cbuf().copy(), r.zbuf().copy()
    finally:
        r.set_clip_window(0, 0, W - 1, H - 1)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def run_mirror(L, texs, steps, whole_objects):
    keep = []
    for h, (tex, lg) in enumerate(texs):
        L.lte_set_texture(h, _p(tex), lg)
    L.lte_init_clipping(0, 0, W - 1, H - 1)
    for st in steps:
        if st[0] == "clear":
            L.lte_clear(st[1], 255, 255, 255)
        elif st[0] == "window":
            L.lte_init_clipping(*st[1])
        elif st[0] == "gpolys":
            for (e, t, col, tex) in st[2]:
                tt = np.ascontiguousarray(t, dtype=np.int32)
                keep.append(tt)
                L.lte_g_polygon(st[1], _p(tt), e, col, 0, 0, 0, tex)
        elif st[0] == "prp":
            for (e, t, O, u, v, tex, ls) in st[1]:
                a = [np.ascontiguousarray(x, dtype=np.int32) for x in (t, O, u, v)]
                keep += a
                L.lte_g_prp_polygon(_p(a[0]), e, _p(a[1]), _p(a[2]), _p(a[3]), tex, ls, 7)
        elif st[0] == "object":
            _, sc, rt, obj = st
            v, po, pa, pv = sc.verts.copy(), sc.polys.copy(), sc.polyaux.copy(), sc.polyverts.copy()
            keep += [v, po, pa, pv]
            if whole_objects:
                L.lte_render_object(_p(v), v.shape[0], _p(po), _p(pa), _p(pv), po.shape[0], rt, 0, 0, 0, 0, 0, 0, sc.log_focus, *obj[3:], *obj[:3])
            else:
                L.lte_render_polys_one_by_one(_p(v), v.shape[0], _p(po), _p(pa), _p(pv), po.shape[0], rt, sc.log_focus, *obj[3:], *obj[:3])
        else:
            for l in st[1]:
                a, b = np.ascontiguousarray(l[0:3]), np.ascontiguousarray(l[3:6])
                if l[7] == 0:
                    L.lte_g_line(_p(a), _p(b), int(l[6]), 255, 255, 255)
                elif l[7] == 1:
                    L.lte_g_dot(_p(a), int(l[6]), 255, 255, 255)
                else:
                    L.lte_g_pixel(_p(a), int(l[6]))
    L.lte_flush()
    c = np.ctypeslib.as_array(L.lte_cbuf(), shape=(H, W)).copy()
    z = np.ctypeslib.as_array(L.lte_zbuf(), shape=(H, W)).copy()
    L.lte_init_clipping(0, 0, W - 1, H - 1)
    return c, z


def reference_frames(texs, steps, path):
    """in a child process: the frames, or nothing if the reference stopped"""
    pid = os.fork()
    if pid == 0:
        try:
            c, z = run_reference(R.Ref(W, H), texs, steps)
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
    texs, steps = make_case(it)
    cases.append((texs, steps))
    refs.append(reference_frames(texs, steps, os.path.join(tmp, f"{it}.npz")))
    if it % 50 == 49:
        print("reference frames", it + 1, f"{time.time() - t0:.0f} s", flush=True)
print("reference stopped on", sum(r is None for r in refs), "of", N, "cases", flush=True)

import torch  # noqa: E402,F401  (the GPU from here on)
from raytracing_amd import _lib  # noqa: E402
_lib.lib()
L = C.CDLL(os.path.join(os.path.dirname(_lib.SO_PATH), "liblt_engine.so"))
L.lte_cbuf.restype = C.POINTER(C.c_int32)
L.lte_zbuf.restype = C.POINTER(C.c_int32)
L.lte_init(W, H)
L.lte_track_clipper_stores(1)
bad = total = 0
for it, ((texs, steps), ref) in enumerate(zip(cases, refs)):
    if ref is None:
        continue
    for whole in (it % 2 == 0, it % 2 == 1):
        c, z = run_mirror(L, texs, steps, whole)
        total += 1
        if not (np.array_equal(c, ref[0]) and np.array_equal(z, ref[1])):
            bad += 1
            print("MISMATCH case", it, "whole objects" if whole else "polygon by polygon", [s[0] + (":%d" % s[2] if s[0] == "object" else "") for s in steps],
                  "c px", int((c != ref[0]).sum()), "z px", int((z != ref[1]).sum()), flush=True)
    if it % 50 == 49:
        print("compared", it + 1, "bad", bad, f"{time.time() - t0:.0f} s", flush=True)
L.lte_track_clipper_stores(0)
print("TOTAL", total, "frames against the reference build's,", bad, "differing;", sum(r is None for r in refs), "cases skipped (reference stopped)")
