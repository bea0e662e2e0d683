#!/usr/bin/env python3
"""Generate the committed golden vectors by RUNNING THE REFERENCE'S OWN CODE.

Runs only where /root/reference is present (the build container): it loads the checker
libraries that oracle/build_ref.sh compiles from the reference sources and records, for
fixed seeded inputs, what the reference computes.  The outputs are data only -- inputs
and expected outputs -- stored as .npz (no pickles) and .json next to this script:

  stage_kats.npz    per-stage known answers (transform, perspective, clippers, edge scan
                    row tables, CL_light)
  frames.json       per-scene digests of the colour and depth buffers + fragment-free
                    metadata; scenes are regenerated from seeds by raytracing_amd.scenes
  c1_scene.npz      configuration 1: the tracer.dat world (quad + three spheres, tessellated
                    and lit by this script) as flat scene arrays, with the reference's frame
  crops.npz         a few raw 32x32 crops of reference frames
  stale_depth_kats.json  frames with screen-clipped lines drawn after filled polygons (their end
                    points' depth is a leftover in the clippers' static stores)
  prp_kats.json     frames of G_prp_textured_polygon calls (the perspective mapper at the G_* level)
  c4_frames.json    configuration 4: digests of thirty-two frames of the animated 1M-triangle batch
  c5_frame.json     configuration 5 in shape: digests of one 16384 x 16384 frame of 1M triangles
  data_kats.json    what the reference's D_data builds from the .dat files of
                    tests/test_data_loader.py and from tracer.dat

Usage:  python tests/golden/make_golden.py
"""
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import oracle.port as P  # noqa: E402
import oracle.ref as R  # noqa: E402
from raytracing_amd import scenes as S  # noqa: E402

REFDIR = "/root/reference"


# ----------------------------------------------------------------------------- frames

def frame_cases():
    """(name, scene factory, clear tuple) -- shared with the tests through frames.json keys."""
    cases = []
    for seed in range(6):
        for rt in (S.M_FLAT, S.M_SHADED, S.M_TEXTURED):
            cases.append((f"fuzz160x120-rt{rt}-s{seed}",
                          dict(gen="fuzz", n=70, w=160, h=120, lf=7, seed=1000 + seed * 7 + rt, rt=rt,
                               textured=rt == S.M_TEXTURED, equal_lights=False)))
    for seed in range(4):
        cases.append((f"fuzzlin160x120-s{seed}",
                      dict(gen="fuzzlin", n=70, w=160, h=120, lf=7, seed=2000 + seed)))
    cases.append(("c2small-640x480-20k", dict(gen="c2", n=20000, w=640, h=480, lf=9, radius=8, seed=0xC0FFEE02)))
    cases.append(("c2large-640x480-2k", dict(gen="c2", n=2000, w=640, h=480, lf=9, radius=96, seed=0xC0FFEE12)))
    cases.append(("c2small-1920x1080-50k", dict(gen="c2", n=50000, w=1920, h=1080, lf=10, radius=8, seed=0xC0FFEE02)))
    cases.append(("c3lin-640x480-20k", dict(gen="c3", n=20000, w=640, h=480, lf=9, radius=10, seed=0xC0FFEE03,
                                            equal_lights=True, lin_percent=100)))
    cases.append(("c3mix-640x480-20k", dict(gen="c3", n=20000, w=640, h=480, lf=9, radius=10, seed=0xC0FFEE03,
                                            equal_lights=True, lin_percent=90)))
    cases.append(("c3q2-640x480-5k", dict(gen="c3", n=5000, w=640, h=480, lf=9, radius=10, seed=0xC0FFEE33,
                                          equal_lights=False, lin_percent=100)))
    return cases


def build_scene(spec):
    g = spec["gen"]
    if g == "fuzz":
        return S.gen_fuzz_polys(spec["n"], spec["w"], spec["h"], spec["lf"], spec["seed"], render_type=spec["rt"],
                                textured=spec["textured"], equal_lights=spec["equal_lights"])
    if g == "fuzzlin":
        sc = S.gen_fuzz_polys(spec["n"], spec["w"], spec["h"], spec["lf"], spec["seed"], render_type=S.M_TEXTURED,
                              textured=True, equal_lights=True, zrange=(1100, 3000))
        return sc
    if g == "c2":
        return S.gen_c2(spec["n"], spec["w"], spec["h"], spec["lf"], radius=spec["radius"], seed=spec["seed"])
    if g == "c3":
        return S.gen_c3(spec["n"], spec["w"], spec["h"], spec["lf"], radius=spec["radius"], seed=spec["seed"],
                        equal_lights=spec["equal_lights"], lin_percent=spec["lin_percent"])
    raise ValueError(g)


CLEAR = (0x102030, 255, 255, 255)


def render_ref(sc):
    ref = R.Ref(sc.width, sc.height)
    for h, (tex, lg) in enumerate(sc.textures):
        ref.set_texture(h, tex, lg)
    ref.clear(*CLEAR)
    vv = ref.transform(sc.matrix, sc.verts)
    ref.render_polys(vv, sc.polys, sc.polyaux, sc.polyverts, sc.render_type, sc.log_focus, sc.force_linear)
    return ref.cbuf().copy(), ref.zbuf().copy()


# ----------------------------------------------------------------------------- stage KATs

def stage_kats():
    ref = R.Ref(160, 120)
    out = {}
    rng = S.Rng(4242)
    # trig tables: both variants of T_init_math
    s, c = ref.trig_tables()
    out["trig_sin_trans"], out["trig_cos_trans"] = s, c
    s2, c2 = R.Ref(160, 120, "transbs").trig_tables()
    out["trig_sin_transbs"], out["trig_cos_transbs"] = s2, c2
    # matrices + transform
    cams = rng.ints(0, 255, 3 * 8).reshape(8, 3)
    camp = rng.ints(-500, 500, 3 * 8).reshape(8, 3)
    obja = rng.ints(0, 255, 3 * 8).reshape(8, 3)
    objp = rng.ints(-2000, 2000, 3 * 8).reshape(8, 3)
    mats = []
    for i in range(8):
        mats.append(ref.make_matrix(tuple(cams[i]) + tuple(camp[i]), tuple(obja[i]) + tuple(objp[i]), 8))
    out["mat_cam"] = np.concatenate([cams, camp], axis=1).astype(np.int32)
    out["mat_obj"] = np.concatenate([obja, objp], axis=1).astype(np.int32)
    out["mat_m12"] = np.array(mats, dtype=np.int32)
    v = rng.ints(-70000, 70000, 4 * 256).reshape(-1, 4).astype(np.int32)
    out["xf_in"] = v
    out["xf_out"] = np.array([ref.transform(m, v) for m in mats], dtype=np.int32)
    # perspective
    t = np.zeros((200, 8), np.int32)
    t[:, 0] = rng.ints(-300000, 300000, 200)
    t[:, 1] = rng.ints(-300000, 300000, 200)
    t[:, 2] = rng.ints(10, 60000, 200)
    t[:100, 0] = rng.ints(-(2 ** 31) + 1, 2 ** 31 - 1, 100)       # needs the 64-bit quotient
    t[:, 3:] = rng.ints(-(2 ** 31), 2 ** 31 - 1, 200 * 5).reshape(200, 5)
    out["persp_in"] = t
    out["persp_out_lf8"] = ref.perspective(t, 8, 8)
    out["persp_out_lf11"] = ref.perspective(t, 8, 11)
    # volume clipping
    mn = rng.ints(-3000, 3000, 3 * 300).reshape(-1, 3).astype(np.int32)
    ext = rng.ints(0, 3000, 3 * 300).reshape(-1, 3).astype(np.int32)
    mx = mn + ext
    mn[:40, 2] = rng.ints(-5, 15, 40)
    mx[:40, 2] = mn[:40, 2] + rng.ints(0, 6, 40)
    out["vol_min"], out["vol_max"] = mn, mx
    out["vol_out"] = np.array([ref.volume_clipping(mn[i], mx[i]) for i in range(300)], np.int32)
    # back-face test
    tri = rng.ints(-40000, 40000, 6 * 300).reshape(-1, 3, 2).astype(np.int32)
    tri[:20, 2] = tri[:20, 0]                                      # degenerate: zero area
    out["cull_in"] = tri
    out["cull_out"] = np.array([ref.normal_z_negative(tri[i, 0], tri[i, 1], tri[i, 2]) for i in range(300)], np.int32)
    # line clippers: axis 0 (x, bounds 0..159), 1 (y, 0..119), 2 (z, >=10)
    for axis, name in ((0, "x"), (1, "y"), (2, "z")):
        a = rng.ints(-400, 600, 8 * 400).reshape(-1, 8).astype(np.int32)
        b = rng.ints(-400, 600, 8 * 400).reshape(-1, 8).astype(np.int32)
        a[:, 3:] = rng.ints(-(2 ** 24), 2 ** 24, 5 * 400).reshape(-1, 5)
        b[:, 3:] = rng.ints(-(2 ** 24), 2 ** 24, 5 * 400).reshape(-1, 5)
        # exact-bound and equal-coordinate cases
        bound_lo = 10 if axis == 2 else 0
        bound_hi = {0: 159, 1: 119, 2: 0}[axis]
        a[:30, axis] = bound_lo
        b[30:60, axis] = bound_hi
        b[60:80, axis] = a[60:80, axis]
        a[80:100, axis] = bound_lo - 1
        b[100:120, axis] = bound_hi + 1
        res = np.zeros((400, 2 + 16), np.int32)
        for i in range(400):
            r, o1, o2, moved = ref.line_clipping(axis, a[i], b[i])
            res[i, 0], res[i, 1] = r, moved
            if r:
                res[i, 2:10], res[i, 10:18] = o1, o2
        out[f"line{name}_a"], out[f"line{name}_b"], out[f"line{name}_out"] = a, b, res
    # polygon clippers + edge scan row tables on random closed polygons (dim 6)
    polys = S.gen_screen_polys(60, 160, 120, seed=77, dim=6, max_edges=6)
    pin, pn, pout, pon = [], [], [], []
    scan = []
    for (e, t, col) in polys:
        n, o = ref.polygon_clipping(0, t, 6)
        pad = np.zeros((17, 6), np.int32)
        pad[: e + 1] = t
        pin.append(pad)
        pn.append(e)
        po = np.zeros((17, 6), np.int32)
        po[: min(n + 1, 17)] = o[: min(n + 1, 17)]
        pout.append(po)
        pon.append(n)
        miny, maxy, xs, xe, vs, ve = ref.scan_polygon(t, 6)
        scan.append(np.concatenate([[miny, maxy], xs, xe, vs[:, :4].reshape(-1), ve[:, :4].reshape(-1)]).astype(np.int64))
    out["polyx_in"], out["polyx_n"] = np.array(pin), np.array(pn, np.int32)
    out["polyx_out"], out["polyx_on"] = np.array(pout), np.array(pon, np.int32)
    out["scan_tables"] = np.array(scan, dtype=np.int64)
    # z polygon clipper on view-space polygons (dim 6)
    zin, zn, zout, zon = [], [], [], []
    for i in range(60):
        e = int(rng.ints(3, 6, 1)[0])
        t = np.zeros((e + 1, 6), np.int32)
        t[:e, 0] = rng.ints(-500, 500, e)
        t[:e, 1] = rng.ints(-500, 500, e)
        t[:e, 2] = rng.ints(-40, 80, e)
        t[:e, 3:] = rng.ints(0, 255 << 16, 3 * e).reshape(e, 3)
        t[e] = t[0]
        n, o = ref.polygon_clipping(2, t, 6)
        pad = np.zeros((17, 6), np.int32)
        pad[: e + 1] = t
        zin.append(pad)
        zn.append(e)
        po = np.zeros((17, 6), np.int32)
        po[: min(n + 1, 17)] = o[: min(n + 1, 17)]
        zout.append(po)
        zon.append(n)
    out["polyz_in"], out["polyz_n"] = np.array(zin), np.array(zn, np.int32)
    out["polyz_out"], out["polyz_on"] = np.array(zout), np.array(zon, np.int32)
    # CL_light / CL_colour
    cl = rng.ints(0, 0xFFFFFF, 500).astype(np.int32)
    lr = rng.ints(0, 400, 3 * 500).reshape(-1, 3).astype(np.int32)
    out["cl_in_colour"], out["cl_in_light"] = cl, lr
    out["cl_light_out"] = np.array([ref.cl_light(int(cl[i]), *map(int, lr[i])) for i in range(500)], np.int32)
    out["cl_colour_out"] = np.array([ref.cl_colour(*map(int, lr[i])) for i in range(500)], np.int32)
    return out


# ----------------------------------------------------------------------------- configuration 1

def parse_tracer_dat(path):
    """The few numbers of the reference's tracer.dat world: lights, spheres, one polygon."""
    txt = open(path).read()
    num = r"[-+]?\d*\.?\d+"
    lights = [tuple(map(float, m)) for m in re.findall(
        rf"var light \w+ \{{\[({num}) ({num}) ({num})\] \[({num}) ({num}) ({num})\]\}}", txt)]
    spheres = [tuple(map(float, m)) for m in re.findall(
        rf"var sphere \w+ \{{ \d+ \{{\[({num}) ({num}) ({num})\].*?\}}\s*\[\s*({num})\s+({num})\s+({num})\]\s*({num})\s*\}}", txt)]
    vrt = re.search(r"var \[15\]float vrt \[(.*?)\]", txt, re.S).group(1).split()
    quad = np.array(list(map(float, vrt)), dtype=np.float64).reshape(5, 3)
    pcol = tuple(map(float, re.search(rf"var polygon \w+\s+\{{ \d+ \{{\[({num}) ({num}) ({num})\]", txt).groups()))
    return lights, spheres, quad, pcol


def c1_scene():
    """tracer.dat as polygon objects (raytracing_amd.scenes.tracer_polygon_objects)."""
    return S.tracer_polygon_objects(*parse_tracer_dat(os.path.join(REFDIR, "tracer.dat")))


def c1_golden():
    W, H, LF = 640, 480, 9
    cam = (0, 0, 0, 0, -60, -700)
    objs = c1_scene()
    ref = R.Ref(W, H)
    ref.clear(0x000000, 0, 0, 0)
    out = {}
    for i, o in enumerate(objs):
        m = ref.make_matrix(cam, (0, 0, 0, 0, 0, 0), LF)     # also sets the reference's camera
        ref.render_object(o["verts"], o["polys"], None, o["polyverts"], S.M_SHADED, (0, 0, 0), (0, 0, 0))
        out[f"obj{i}_verts"], out[f"obj{i}_polys"], out[f"obj{i}_polyverts"] = o["verts"], o["polys"], o["polyverts"]
        out[f"obj{i}_m12"] = m
    out["n_objs"] = np.array([len(objs)], np.int32)
    out["meta"] = np.array([W, H, LF], np.int32)
    out["colour"] = ref.cbuf().copy()
    out["depth"] = ref.zbuf().copy()
    return out


def data_kats():
    """The reference's D_data (data.cpp:676-781) on the two files of tests/test_data_loader.py and on tracer.dat."""
    import ctypes as C
    import tempfile
    sys.path.insert(0, os.path.dirname(HERE))
    import test_data_loader as T
    ref = T.ref_lib()
    d = tempfile.mkdtemp()
    T.write_files(d)
    cwd = os.getcwd()
    os.chdir(d)
    try:
        out = {"main": T.dump_main(ref.ltref_d_data(b"main.dat")),
               "tracer": T.dump_world(ref.ltref_d_data(os.path.join(REFDIR, "tracer.dat").encode()))}
    finally:
        os.chdir(cwd)
    return out


def shade_kats():
    """Shading pre-pass vectors: the reference's M_shade_polygon_object on generated cases."""
    r = R.Ref(160, 120)
    out = {}
    cases = [(120, 7001, (0, 0, 0), (0, 0, 0), True), (200, 7002, (13, 200, 77), (100, -50, 400), True),
             (90, 7003, (250, 3, 128), (-2000, 900, 15), False)]
    for k, (n, seed, ang, pos, far) in enumerate(cases):
        c = S.gen_shading_case(n, seed, far_light=far)
        pv, pa = r.shade_object(c["verts"], c["normals"], c["polys"], c["polyverts"], c["polynormals"], None, pos, ang,
                                c["lights"])
        out[f"np_{k}"] = n
        out[f"seed_{k}"] = seed
        out[f"ang_{k}"] = np.array(ang, np.int32)
        out[f"pos_{k}"] = np.array(pos, np.int32)
        out[f"far_{k}"] = far
        out[f"s9_{k}"] = r.self_matrix(*ang)
        out[f"rgb_{k}"] = pv[:, 1:4].copy()
        out[f"flat_{k}"] = pa[:, :3].copy()
    out["n_cases"] = len(cases)
    return out


NORMAL_CASES = [dict(n_long=16, n_lat=8, radius=300, centre=(0, 0, 0), mix=0),
                dict(n_long=12, n_lat=6, radius=2500, centre=(40, -900, 7000), mix=1),
                dict(n_long=20, n_lat=10, radius=90, centre=(-5, 3, 200), mix=2)]


def normal_case_types(types, mix):
    """m_type per polygon of a normals case: the generator's, or with planar|curved, typeless
    and planar polygons mixed in"""
    ty = types.copy()
    if mix >= 1:
        ty[::5] = 3
        ty[1::7] = 0
    if mix >= 2:
        ty[2::3] = 1
    return ty


def normals_kats():
    """Object set-up vectors: the reference's M_init_polygon_object on generated spheres."""
    r = R.Ref(160, 120)
    out = {}
    for k, c in enumerate(NORMAL_CASES):
        ob = S.gen_sphere_object(c["n_long"], c["n_lat"], c["radius"], c["centre"])
        ty = normal_case_types(ob["types"], c["mix"])
        n, pn = r.init_object(ob["verts"], ob["polys"], ty, ob["polyverts"])
        out[f"normals_{k}"] = n
        out[f"polynormals_{k}"] = pn
    out["n_cases"] = len(NORMAL_CASES)
    return out


WIRE_W, WIRE_H = 200, 150


def wire_lines(seed, n, lo, hi, zlo, zhi, w=WIRE_W, h=WIRE_H):
    """n lines {x1 y1 z1 x2 y2 z2 lit-colour 0}: end points uniform in [lo, hi] relative to the
    screen (x in [lo, w + hi], y in [lo, h + hi]), depths in [zlo, zhi]"""
    rng = S.Rng(seed)
    l = np.zeros((n, 8), np.int32)
    l[:, 0] = rng.ints(lo, w + hi, n); l[:, 1] = rng.ints(lo, h + hi, n)
    l[:, 3] = rng.ints(lo, w + hi, n); l[:, 4] = rng.ints(lo, h + hi, n)
    l[:, 2] = rng.ints(zlo, zhi, n); l[:, 5] = rng.ints(zlo, zhi, n)
    l[:, 6] = rng.ints(1, 0xFFFFFF, n)
    return l


def wire_cases():
    """name -> list of steps; a step is ("clear", colour) | ("object", scene, render_type, obj) |
    ("lines", int32[n][8]).  Clipped lines appear only on frames without filled polygons
    (their depth is then immaterial: nothing has written the depth buffer)."""
    cases = {}
    for k in range(3):
        sc = S.gen_fuzz_polys(150, WIRE_W, WIRE_H, 7, 5000 + k, render_type=S.M_FLAT)
        cases[f"wire_object_{k}"] = [("clear", 0x102030), ("object", sc, S.M_WIRE, (3 + k, 4, 5, 0, 0, 30))]
    cases["lines_clipped"] = [("clear", 0), ("lines", wire_lines(77, 400, -150, 200, -1000, 1000))]
    cases["lines_degenerate"] = [("clear", 0x000040), ("lines", np.array(
        [[5, 5, 0, 5, 5, 0, 0x111111, 0], [0, 0, 1, WIRE_W - 1, WIRE_H - 1, 2, 0x222222, 0],
         [WIRE_W - 1, 0, 3, 0, WIRE_H - 1, 3, 0x333333, 0], [-50, 70, 0, 400, 70, 0, 0x444444, 0],
         [60, -90, 0, 60, 900, 0, 0x555555, 0], [0, 10, 0, 0, 140, 0, 0x666666, 0],
         [WIRE_W - 1, 10, 0, WIRE_W - 1, 140, 0, 0x777777, 0], [-5, -5, 0, -1, -1, 0, 0x888888, 0],
         [WIRE_W, 3, 0, WIRE_W + 40, 100, 0, 0x999999, 0], [10, 0, 0, 150, 0, 0, 0xaaaaaa, 0],
         [10, WIRE_H - 1, 0, 150, WIRE_H - 1, 0, 0xbbbbbb, 0]], np.int32))]
    pts = wire_lines(80, 600, -10, 10, -6200, -100)
    pts[:, 7] = 1 + (np.arange(600) % 2)                        # G_dot and G_pixel in turn
    pts[pts[:, 7] == 2, 0] = np.clip(pts[pts[:, 7] == 2, 0], 0, WIRE_W - 1)      # G_pixel is not bounds-checked
    pts[pts[:, 7] == 2, 1] = np.clip(pts[pts[:, 7] == 2, 1], 0, WIRE_H - 1)
    cases["dots_pixels"] = [("clear", 0), ("object", S.gen_c2(600, WIRE_W, WIRE_H, 7, radius=25, seed=8), S.M_SHADED, (0, 0, 0, 0, 0, 0)),
                            ("lines", pts)]
    sc = S.gen_c2(800, WIRE_W, WIRE_H, 7, radius=20, seed=5)
    cases["mixed_unclipped"] = [("clear", 0), ("object", sc, S.M_SHADED, (0, 0, 0, 0, 0, 0)),
                                ("lines", wire_lines(78, 300, 1, -2, -6200, -100)),
                                ("object", S.gen_c2(300, WIRE_W, WIRE_H, 7, radius=25, seed=6), S.M_SHADED, (0, 0, 0, 0, 0, 0)),
                                ("lines", wire_lines(79, 100, 1, -2, -6200, -100))]
    return cases


def gate_inputs(n=200, seed=5, w=WIRE_W, h=WIRE_H):
    """n portal polygons in view space with the window each is applied to:
    list of (indices int32[e+1], verts int32[e][4], start window int32[4])"""
    rng = S.Rng(seed)
    out = []
    for _ in range(n):
        e = int(rng.ints(3, 6, 1)[0])
        v = np.zeros((e, 4), np.int32)
        v[:, :2] = rng.ints(-400, 400, e * 2).reshape(-1, 2)
        v[:, 2] = rng.ints(-50, 900, e)
        idx = np.array(list(range(e)) + [0], np.int32)
        start = np.array([int(rng.ints(0, 60, 1)[0]), int(rng.ints(0, 40, 1)[0]), int(rng.ints(120, w - 1, 1)[0]),
                          int(rng.ints(90, h - 1, 1)[0])], np.int32)
        out.append((idx, v, start))
    return out


def gate_kats():
    """M_set_gate of the reference build on gate_inputs(): return values and windows left behind."""
    r = R.Ref(WIRE_W, WIRE_H)
    res = [r.set_gate(idx, v, 7, start) for (idx, v, start) in gate_inputs()]
    return dict(rc=np.array([a for a, _ in res], np.int32), bounds=np.array([b for _, b in res], np.int32))


BICUBIC_LIGHTS = np.array([[1, 0, 0, 0, 0, 30, 30, 30], [2, 400, -500, -300, 0, 220, 200, 180],
                           [4, 40, -90, 80, 0, 90, 100, 110]], np.int32)


def bicubic_cases():
    """(name, controls, log_render_size, render type, position, angles) of single patches drawn at
    200x150 with log_focus 7, texture space 2^7, texture make_texture(9, 6) for M_TEXTURED"""
    out = []
    for seed in range(6):
        c = S.gen_bicubic_patch(seed, extent=250 + 40 * seed, bulge=100 + 30 * seed, centre=(10 * seed, -20, 300 + 100 * seed))
        rt = (S.M_FLAT, S.M_SHADED, S.M_TEXTURED, S.M_WIRE, S.M_SHADED, S.M_TEXTURED)[seed]
        out.append((f"patch_{seed}", c, 2 + seed % 3, rt, (5, -3, 40 * seed), (seed * 7, 250 - seed * 3, seed)))
    return out


def run_bicubic_case(case, be):
    """be: oracle.port.Oracle or oracle.ref.Ref of WIRE_W x WIRE_H -> (normals, intensities, colour, depth)"""
    name, c, lrs, rt, pos, ang = case
    be.set_texture(0, S.make_texture(9, 6), 6)
    be.clear(0x000020, 255, 255, 255)
    n, i = be.bicubic(c, lrs, 0xFFFFFF, 7, 0 if rt == S.M_TEXTURED else -1, pos, ang, BICUBIC_LIGHTS, rt, 7, 1)
    return n[:, :3].copy(), i, be.cbuf().copy(), be.zbuf().copy()


def bicubic_kats():
    """Patches set up, shaded and drawn by the reference build: normals, intensities, frame digests."""
    out = {}
    for case in bicubic_cases():
        n, i, c, z = run_bicubic_case(case, R.Ref(WIRE_W, WIRE_H))
        out[case[0] + "_normals"] = n
        out[case[0] + "_intensities"] = i
        out[case[0] + "_digests"] = np.array([S.frame_digest(c), S.frame_digest(z)])
    return out


def window_cases():
    """Frames drawn under clip windows smaller than the screen (C_init_clipping): steps as in
    wire_cases plus ("window", (minx, miny, maxx, maxy)) and ("gpolys", kind, polygons of
    gen_screen_polys).  Lines that a window clips come before any filled polygon."""
    w, h = WIRE_W, WIRE_H
    cases = {}
    for k, rt in enumerate((S.M_FLAT, S.M_SHADED, S.M_TEXTURED)):
        sc = S.gen_fuzz_polys(160, w, h, 7, 6000 + k, render_type=rt, textured=rt == S.M_TEXTURED, equal_lights=True)
        cases[f"object_{rt}"] = [("clear", 0x101010), ("window", (30 + k, 20, 170, 120 - k)), ("object", sc, rt, (2, 3, 4, 0, 0, 20))]
    cases["screen_polys"] = [("clear", 0), ("window", (50, 40, 150, 110)), ("gpolys", 4, S.gen_screen_polys(80, w, h, seed=61, dim=6))]
    a = S.gen_c2(700, w, h, 7, radius=25, seed=62)
    b = S.gen_c2(700, w, h, 7, radius=25, seed=63)
    cases["two_windows"] = [("clear", 0x000030), ("window", (0, 0, 99, h - 1)), ("lines", wire_lines(64, 150, -80, 80, -500, 500)),
                            ("object", a, S.M_SHADED, (0, 0, 0, 0, 0, 0)),
                            ("window", (100, 10, w - 1, 100)), ("object", b, S.M_SHADED, (0, 0, 0, 0, 0, 0)),
                            ("window", (0, 0, w - 1, h - 1)), ("lines", wire_lines(65, 80, 1, -2, -6200, -100))]
    sc = S.gen_fuzz_polys(150, w, h, 7, 6007, render_type=S.M_FLAT)
    cases["wire_object"] = [("clear", 0), ("window", (25, 25, 120, 140)), ("object", sc, S.M_WIRE, (3, 4, 5, 0, 0, 30))]
    return cases


def run_wire_case(steps, be):
    """be: oracle.port.Oracle or oracle.ref.Ref of WIRE_W x WIRE_H"""
    try:
        return _run_steps(steps, be)
    finally:
        be.set_clip_window(0, 0, be.w - 1, be.h - 1)        # a process-wide global in the reference build


def _run_steps(steps, be):
    for st in steps:
        if st[0] == "clear":
            be.clear(st[1], 255, 255, 255)
        elif st[0] == "window":
            be.set_clip_window(*st[1])
        elif st[0] == "gpolys":
            for (e, t, col) in st[2]:
                be.g_polygon(st[1], t, colour=col)
        elif st[0] == "object":
            _, sc, rt, obj = st
            m = P.make_matrix((0, 0, 0, 0, 0, 0), obj, 0)
            be.render_polys(P.transform(m, sc.verts), sc.polys, sc.polyaux, sc.polyverts, rt, sc.log_focus)
        else:
            for l in st[1]:
                if l[7] == 0:
                    be.g_line(l[0:3], l[3:6], int(l[6]), 255, 255, 255)
                elif l[7] == 1:
                    be.g_dot(l[0:3], int(l[6]), 255, 255, 255)
                else:
                    be.g_pixel(l[0:3], int(l[6]))
    return be.cbuf().copy(), be.zbuf().copy()


def _textures_of(steps):
    for st in steps:
        if st[0] == "object" and st[1].textures:
            return st[1].textures
    return []


def window_kats():
    """Frames under smaller clip windows drawn by the reference build: digests."""
    out = {}
    for name, steps in window_cases().items():
        r = R.Ref(WIRE_W, WIRE_H)
        for hnd, (tex, lg) in enumerate(_textures_of(steps)):
            r.set_texture(hnd, tex, lg)
        c, z = run_wire_case(steps, r)
        out[name] = dict(colour=S.frame_digest(c), depth=S.frame_digest(z), drawn=int((c != c[0, 0]).sum()))
    return out


def wire_kats():
    """Wireframe frames (M_WIRE objects, G_line lists) drawn by the reference build: digests."""
    out = {}
    for name, steps in wire_cases().items():
        c, z = run_wire_case(steps, R.Ref(WIRE_W, WIRE_H))
        out[name] = dict(colour=S.frame_digest(c), depth=S.frame_digest(z), drawn=int((c != c[0, 0]).sum()))
    return out


def bench_frame():
    """The benchmark's own frame (1M Gouraud triangles at 1080p), drawn by the reference build."""
    sc = S.gen_c2(1_000_000, 1920, 1080, 10, radius=8)
    ref = R.Ref(sc.width, sc.height)
    ref.clear(0, 0, 0, 0)
    ref.render_polys(P.transform(sc.matrix, sc.verts), sc.polys, sc.polyaux, sc.polyverts, sc.render_type, sc.log_focus)
    c, z = ref.cbuf().copy(), ref.zbuf().copy()
    return dict(workload="c2-small: gen_c2(1000000, 1920, 1080, 10, radius=8), clear 0", colour=S.frame_digest(c),
                depth=S.frame_digest(z), drawn_pixels=int((z != 2 ** 31 - 1).sum()))


def stale_depth_cases():
    """Frames in which lines that the screen clips are drawn AFTER filled polygons: G_line clips with
    dimension 2 (GraphicsBase.cpp:191-193), so a clipped end point's depth is what the polygon
    clips before it left in the clippers' static stores (Clipp2D.cpp:99-104, 172-177).  Every case
    draws, right after its clear, one quadrilateral that reaches beyond all four sides of the
    screen: its clips write every one of the twelve stores (its top and bottom edges both x sides,
    the x-clipped outline's two upright edges both y sides), so the frames do not depend on what
    ran before the case.  Steps as in wire_cases."""
    w, h = WIRE_W, WIRE_H
    cases = {}

    def overwrite_stores(k):
        t = np.array([[-w - 7 * k, -h - 3, -11 - k], [2 * w + 5, -h - 5 * k, -23], [2 * w + 9 * k, 2 * h + 1, -37 - 2 * k],
                      [-w - 1, 2 * h + 11 * k, -5], [0, 0, 0]], dtype=np.int32)
        t[4] = t[0]
        return ("gpolys", S.M_FLAT, [(4, t, 0x203040 + 0x050301 * k)])
    a = S.gen_fuzz_polys(180, w, h, 7, 9100, render_type=S.M_SHADED)
    b = S.gen_fuzz_polys(120, w, h, 7, 9101, render_type=S.M_FLAT)
    cases["shaded_flat_lines"] = [("clear", 0x101820), overwrite_stores(8), ("object", a, S.M_SHADED, (3, 4, 5, 0, 0, 30)),
                                  ("lines", wire_lines(9110, 300, -150, 200, -30000, -300)),
                                  ("object", b, S.M_FLAT, (250, 2, 9, 4, -3, 25)),
                                  ("lines", wire_lines(9111, 200, -90, 120, -30000, -300))]
    t = S.gen_fuzz_polys(160, w, h, 7, 9102, render_type=S.M_TEXTURED, textured=True, equal_lights=True, zrange=(12, 2000))
    wsc = S.gen_fuzz_polys(140, w, h, 7, 9103, render_type=S.M_FLAT)
    cases["textured_wire_object"] = [("clear", 0x000030), overwrite_stores(9), ("object", t, S.M_TEXTURED, (1, 2, 3, 0, 0, 20)),
                                     ("object", wsc, S.M_WIRE, (3, 4, 5, 0, 0, 30)),
                                     ("lines", wire_lines(9112, 250, -200, 250, -40000, -200)),
                                     ("object", a, S.M_SHADED, (9, 250, 3, -5, 2, 40)),
                                     ("object", wsc, S.M_WIRE, (250, 7, 1, 6, 0, 12))]
    g = S.gen_screen_polys(90, w, h, seed=9104, dim=6)
    cases["screen_polys_lines"] = [("clear", 0), overwrite_stores(10), ("object", a, S.M_SHADED, (3, 4, 5, 0, 0, 30)), ("gpolys", 4, g),
                                   ("lines", wire_lines(9113, 300, -120, 150, -50000, 50))]
    # more of the same, other seeds: objects of every render type and line lists in turn
    for k in range(8):
        rt = (S.M_SHADED, S.M_FLAT, S.M_TEXTURED)[k % 3]
        o1 = S.gen_fuzz_polys(150 + 10 * k, w, h, 7, 9200 + k, render_type=rt, textured=rt == S.M_TEXTURED, equal_lights=True,
                              zrange=(12, 1500))
        o2 = S.gen_fuzz_polys(60 + 15 * k, w, h, 7, 9300 + k, render_type=S.M_SHADED, max_edges=3 + k % 5)
        steps = [("clear", 0x080808 * k), overwrite_stores(k), ("object", o1, rt, (k, 2 * k, 3 * k, k - 4, 3 - k, 20 + 3 * k)),
                 ("lines", wire_lines(9400 + k, 150 + 20 * k, -160, 220, -45000, -100))]
        if k % 2:
            steps += [("object", o2, S.M_WIRE, (250 - k, 5, k, 2, -2, 15 + k)), ("object", o2, S.M_SHADED, (3, 250, 2 * k, 0, 1, 35)),
                      ("lines", wire_lines(9500 + k, 120, -60, 140, -20000, -500))]
        else:
            steps += [("object", o2, S.M_SHADED, (k, 9, 250, -1, 2, 28)), ("object", o1, S.M_WIRE, (1, k, 5, 3, 0, 22))]
        cases[f"mix_{k}"] = steps
    return cases


def stale_depth_kats():
    """stale_depth_cases() drawn by the reference build: digests."""
    def draw(names):
        out = {}
        for name in names:
            steps = cases[name]
            r = R.Ref(WIRE_W, WIRE_H)
            for hnd, (tex, lg) in enumerate(_textures_of(steps)):
                r.set_texture(hnd, tex, lg)
            c, z = run_wire_case(steps, r)
            out[name] = dict(colour=S.frame_digest(c), depth=S.frame_digest(z), drawn=int((c != c[0, 0]).sum()))
        return out

    cases = stale_depth_cases()
    out = draw(list(cases))
    # the stores are process-wide: a case whose frame depended on the cases before it would
    # come out differently in another order
    again = draw(list(cases)[::-1])
    assert all(again[n] == out[n] for n in cases), [n for n in cases if again[n] != out[n]]
    return out


def prp_cases():
    """Frames of G_prp_textured_polygon calls at WIRE_W x WIRE_H, log_focus 7, texture
    make_texture(9, 6): name -> list of (length, tuples int32[length+1][8], O int32[8], u int32[3],
    v int32[3], texture handle, log_texture_scale).  O, u, v are random (the arithmetic wraps; a
    zero divisor takes the reference's own zero branch)."""
    cases = {}
    for k, equal in enumerate((True, False, True)):
        rng = S.Rng(8100 + k)
        polys = []
        for (e, t, col) in S.gen_screen_polys(70, WIRE_W, WIRE_H, seed=8200 + k, dim=8, max_edges=6, spread=1 + k % 2):
            if equal:
                t[:, 3:6] = t[0, 3:6]
            O = np.zeros(8, np.int32)
            O[0:2] = rng.ints(-400, 400, 2)
            O[2] = rng.ints(20, 1500, 1)[0]
            O[6:8] = rng.ints(0, 1023, 2) << 16
            mag = (3, 40, 300)[k]
            u = rng.ints(-mag, mag, 3).astype(np.int32)
            v = rng.ints(-mag, mag, 3).astype(np.int32)
            if int(rng.ints(0, 9, 1)[0]) == 0:
                v = u.copy()                              # degenerate orientation: zi = zj = zc = 0
            polys.append((e, t, O, u, v, 0, int(rng.ints(4, 9, 1)[0])))
        cases[f"prp_{'equal' if equal else 'vary'}_{k}"] = polys
    return cases


def run_prp_case(polys, be):
    """be: oracle.port.Oracle or oracle.ref.Ref of WIRE_W x WIRE_H"""
    be.set_texture(0, S.make_texture(9, 6), 6)
    be.clear(0x000020, 255, 255, 255)
    for (e, t, O, u, v, tex, ls) in polys:
        be.g_prp_polygon(t, O, u, v, tex, ls, 7)
    return be.cbuf().copy(), be.zbuf().copy()


def prp_kats():
    """G_prp_textured_polygon of the reference build on prp_cases(): digests."""
    out = {}
    for name, polys in prp_cases().items():
        c, z = run_prp_case(polys, R.Ref(WIRE_W, WIRE_H))
        out[name] = dict(colour=S.frame_digest(c), depth=S.frame_digest(z), drawn=int((z != 2 ** 31 - 1).sum()))
    return out


# the four frames of the first fixture and twenty-eight more spread over the 1024-frame batch
C4_FRAMES = tuple(sorted(set((1, 37, 100, 203) + tuple(range(5, 1024, 37)))))


def c4_frames():
    """Configuration 4: frames of the animated batch -- the benchmark's 1M-triangle object under
    frames.frame_matrix(f) (T_set_self_rotation + T_cancatinate_self_world of the reference
    build), cleared per frame -- drawn by the reference build."""
    from raytracing_amd import frames as F
    sc = S.gen_c2(1_000_000, 1920, 1080, 10, radius=8)
    ref = R.Ref(sc.width, sc.height)
    out = {"workload": "c4: gen_c2(1000000, 1920, 1080, 10, radius=8) under frame_matrix(f), clear 0", "frames": {}}
    for f in C4_FRAMES:
        m = ref.make_matrix((0, 0, 0, 0, 0, 0), (f & 255, (2 * f) & 255, (3 * f) & 255, 0, 0, 2048), sc.log_focus)
        assert np.array_equal(m, F.frame_matrix(f)), f
        ref.clear(0, 0, 0, 0)
        ref.render_polys(ref.transform(m, sc.verts), sc.polys, sc.polyaux, sc.polyverts, sc.render_type, sc.log_focus)
        c, z = ref.cbuf(), ref.zbuf()
        out["frames"][str(f)] = dict(colour=S.frame_digest(c), depth=S.frame_digest(z), drawn_pixels=int((z != 2 ** 31 - 1).sum()))
        print("c4 frame", f, out["frames"][str(f)], flush=True)
    return out


C5_TRIS = 1_000_000


def c5_frame():
    """Configuration 5 in shape: one 16384 x 16384 frame, log_focus 13, vertex offsets of +-16
    pixels, at 1M triangles (the reference resets 2 x 16384 table rows per polygon), drawn by a
    16384 x 16384 reference build."""
    sc = S.gen_c2(C5_TRIS, 16384, 16384, 13, radius=16, seed=0xC0FFEE05)
    ref = R.Ref(sc.width, sc.height)
    ref.clear(0, 0, 0, 0)
    ref.render_polys(P.transform(sc.matrix, sc.verts), sc.polys, sc.polyaux, sc.polyverts, sc.render_type, sc.log_focus)
    c, z = ref.cbuf(), ref.zbuf()
    return dict(workload=f"c5 shape: gen_c2({C5_TRIS}, 16384, 16384, 13, radius=16, seed=0xC0FFEE05), clear 0",
                colour=S.frame_digest(c), depth=S.frame_digest(z), drawn_pixels=int((z != 2 ** 31 - 1).sum()))


def main():
    if "--stale" in sys.argv:
        json.dump(stale_depth_kats(), open(os.path.join(HERE, "stale_depth_kats.json"), "w"), indent=1, sort_keys=True)
        return
    if "--prp" in sys.argv:
        json.dump(prp_kats(), open(os.path.join(HERE, "prp_kats.json"), "w"), indent=1, sort_keys=True)
        return
    if "--c4" in sys.argv:
        json.dump(c4_frames(), open(os.path.join(HERE, "c4_frames.json"), "w"), indent=1, sort_keys=True)
        return
    if "--c4c5" in sys.argv:
        json.dump(c4_frames(), open(os.path.join(HERE, "c4_frames.json"), "w"), indent=1, sort_keys=True)
        json.dump(c5_frame(), open(os.path.join(HERE, "c5_frame.json"), "w"), indent=1, sort_keys=True)
        return
    json.dump(bench_frame(), open(os.path.join(HERE, "bench_frame.json"), "w"), indent=1, sort_keys=True)
    json.dump(data_kats(), open(os.path.join(HERE, "data_kats.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(HERE, "shade_kats.npz"), **shade_kats())
    np.savez_compressed(os.path.join(HERE, "normals_kats.npz"), **normals_kats())
    json.dump(wire_kats(), open(os.path.join(HERE, "wire_kats.json"), "w"), indent=1, sort_keys=True)
    json.dump(window_kats(), open(os.path.join(HERE, "window_kats.json"), "w"), indent=1, sort_keys=True)
    json.dump(prp_kats(), open(os.path.join(HERE, "prp_kats.json"), "w"), indent=1, sort_keys=True)
    json.dump(stale_depth_kats(), open(os.path.join(HERE, "stale_depth_kats.json"), "w"), indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "gate_kats.npz"), **gate_kats())
    np.savez_compressed(os.path.join(HERE, "bicubic_kats.npz"), **bicubic_kats())
    if not os.path.isdir(REFDIR):
        raise SystemExit("needs /root/reference (run in the build container)")
    np.savez_compressed(os.path.join(HERE, "stage_kats.npz"), **stage_kats())
    frames, crops = {}, {}
    for name, spec in frame_cases():
        sc = build_scene(spec)
        c, z = render_ref(sc)
        frames[name] = dict(spec=spec, clear=list(CLEAR), colour=S.frame_digest(c), depth=S.frame_digest(z),
                            drawn_pixels=int((z != 2 ** 31 - 1).sum()))
        if name.startswith("c2small-640") or name.startswith("c3mix"):
            crops[name + "_c"] = c[200:232, 300:332].copy()
            crops[name + "_z"] = z[200:232, 300:332].copy()
        print(name, frames[name]["colour"], frames[name]["depth"], frames[name]["drawn_pixels"])
    json.dump(frames, open(os.path.join(HERE, "frames.json"), "w"), indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "crops.npz"), **crops)
    c1 = c1_golden()
    print("c1 drawn pixels", int((c1["depth"] != 2 ** 31 - 1).sum()))
    np.savez_compressed(os.path.join(HERE, "c1_scene.npz"), **c1)
    json.dump(c4_frames(), open(os.path.join(HERE, "c4_frames.json"), "w"), indent=1, sort_keys=True)
    json.dump(c5_frame(), open(os.path.join(HERE, "c5_frame.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
