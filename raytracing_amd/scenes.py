"""Synthetic scene generators for the polygon path (host side, numpy, deterministic).

Scenes use the flat layout the C-ABI takes (include/lt_gpu.h):

  verts      int32[nv][4]   object-space X Y Z (4th lane unused), T_LNG_VECTOR stride
                            (reference Trans.h:27)
  polys      int32[np][4]   {colour, texture handle or -1, no_edges, first}
  polyaux    int32[np][4]   {m_red, m_green, m_blue, m_type | log_texture_space_size<<8} or None
  polyverts  int32[*][6]    M_polygon.m_vertices records [Idx R G B Tx Ty]
                            (reference Engine.h:204-207); polygon i owns records
                            first .. first+no_edges, the closing vertex repeated

All randomness is splitmix64 so that every consumer (reference build, CPU restatement,
HIP path) sees identical integers.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

M_WIRE, M_FLAT, M_SHADED, M_TEXTURED = 1, 2, 4, 8   # reference Engine.h:80-83
M_PLANNAR, M_QUAD_XY = 1, 4                     # reference Engine.h:190-193

_MASK = (1 << 64) - 1


def splitmix64(seed: int, n: int) -> np.ndarray:
    """n successive outputs of splitmix64 started at `seed` (uint64)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed & _MASK) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


class Rng:
    """Stream of bounded integers drawn from splitmix64 in a fixed order."""

    def __init__(self, seed: int):
        self.seed = seed & _MASK
        self.k = 0

    def raw(self, n: int) -> np.ndarray:
        with np.errstate(over="ignore"):
            out = splitmix64((self.seed + self.k * 0x9E3779B97F4A7C15) & _MASK, n)
        self.k += n
        return out

    def ints(self, lo: int, hi: int, n: int) -> np.ndarray:
        """n integers uniform in [lo, hi] (inclusive), int64."""
        span = np.uint64(hi - lo + 1)
        return (self.raw(n) % span).astype(np.int64) + lo


@dataclass
class Scene:
    width: int
    height: int
    log_focus: int
    render_type: int
    verts: np.ndarray
    polys: np.ndarray
    polyverts: np.ndarray
    polyaux: Optional[np.ndarray] = None
    textures: List[tuple] = field(default_factory=list)     # [(texels int32[2^log][2^log], log)]
    matrix: Optional[np.ndarray] = None                      # int32[12]; None = identity
    force_linear: int = 0
    name: str = ""

    @property
    def n_polys(self) -> int:
        return int(self.polys.shape[0])


IDENTITY = np.array([32768, 0, 0, 0, 0, 32768, 0, 0, 0, 0, 32768, 0], dtype=np.int32)


def make_texture(seed: int, log_size: int) -> np.ndarray:
    n = 1 << (2 * log_size)
    return (splitmix64(seed, n) & np.uint64(0x00FFFFFF)).astype(np.int32).reshape(1 << log_size, 1 << log_size)


def _regular_texture_mapping(txy: np.ndarray) -> np.ndarray:
    """Replace the texture coordinates of triangles on which the perspective mapper's
    2x2 fixed-point elimination (reference T_linear_solve, TransBs.cpp:477-505, reached
    through MI_tmapping_vectors, EngBase.cpp:163-175) meets a zero pivot -- the reference
    divides by zero there -- by a fixed well-conditioned triple."""
    t = txy.astype(np.int64)
    r0 = t[:, 1] - t[:, 0]                     # rows of the system: (Tx2-Tx1, Ty2-Ty1), (Tx3-Tx2, Ty3-Ty2)
    r1 = t[:, 2] - t[:, 1]
    swap = np.abs(r1[:, 0]) > np.abs(r0[:, 0])           # pivot search: first strictly larger magnitude
    p = np.where(swap[:, None], r1, r0)
    q = np.where(swap[:, None], r0, r1)
    p0 = p[:, 0]
    safe = np.where(p0 == 0, 1, p0)
    num = p[:, 1] * q[:, 0]
    quo = np.sign(num) * np.sign(safe) * (np.abs(num) // np.abs(safe))      # C division truncates
    q11 = q[:, 1] - quo
    # column 0 all zero: no elimination step, the back substitution then divides by a00 = 0
    bad = (p0 == 0) | (q11 == 0)
    out = txy.copy()
    out[bad] = np.array([[64, 64], [704, 128], [192, 832]], dtype=txy.dtype)
    return out


def _triangles(rng: Rng, n: int, w: int, h: int, lf: int, radius: int, cx_rng, cy_rng, z_rng,
               cull: str, rgb_equal: bool = False, tex: bool = False, first_z_min: int = 0,
               first_z_frac: int = 0, regular_mapping: bool = False):
    """Common body of the triangle generators: unshared vertices, 4 records per triangle."""
    cx = rng.ints(cx_rng[0], cx_rng[1], n)
    cy = rng.ints(cy_rng[0], cy_rng[1], n)
    zc = rng.ints(z_rng[0], z_rng[1], n)
    ox = rng.ints(-radius, radius, 3 * n).reshape(n, 3)
    oy = rng.ints(-radius, radius, 3 * n).reshape(n, 3)
    dz = rng.ints(-16, 16, 3 * n).reshape(n, 3)
    rgb = rng.ints(0, 255, 9 * n).reshape(n, 3, 3)
    flip = rng.ints(0, 1, n)
    txy = rng.ints(0, 1023, 6 * n).reshape(n, 3, 2)
    force = rng.ints(0, 99, n)
    zv = zc[:, None] + dz
    if first_z_min:
        # keep the first vertex at or beyond the linear/perspective mapper switch for
        # first_z_frac percent of the triangles (reference EngBase.cpp:491-493)
        sel = force < first_z_frac
        zv[:, 0] = np.where(sel, np.maximum(zv[:, 0], first_z_min), zv[:, 0])
    px = cx[:, None] + ox
    py = cy[:, None] + oy
    vx = ((px - w // 2) * zv) >> lf
    vy = ((py - h // 2) * zv) >> lf
    # winding: make the projected cross product negative (drawn), then flip per `cull`
    cross = (px[:, 1] - px[:, 0]) * (py[:, 2] - py[:, 0]) - (py[:, 1] - py[:, 0]) * (px[:, 2] - px[:, 0])
    swap = cross > 0
    if cull == "random":
        swap = swap ^ (flip == 1)
    order = np.tile(np.array([0, 1, 2]), (n, 1))
    order[swap] = np.array([0, 2, 1])
    rows = np.arange(n)[:, None]
    vx, vy, zv = vx[rows, order], vy[rows, order], zv[rows, order]
    rgb = rgb[rows, order]
    txy = txy[rows, order]
    if rgb_equal:
        rgb = np.repeat(rgb[:, :1, :], 3, axis=1)
    verts = np.zeros((3 * n, 4), dtype=np.int32)
    verts[:, 0] = vx.reshape(-1)
    verts[:, 1] = vy.reshape(-1)
    verts[:, 2] = zv.reshape(-1)
    pv = np.zeros((n, 4, 6), dtype=np.int32)
    idx = (np.arange(n)[:, None] * 3 + np.arange(3)[None, :])
    pv[:, :3, 0] = idx
    pv[:, :3, 1:4] = rgb
    if tex:
        if regular_mapping:
            txy = _regular_texture_mapping(txy)
        pv[:, :3, 4:6] = txy
    pv[:, 3, :] = pv[:, 0, :]
    polys = np.zeros((n, 4), dtype=np.int32)
    polys[:, 0] = 0x00FFFFFF
    polys[:, 1] = 0 if tex else -1
    polys[:, 2] = 3
    polys[:, 3] = np.arange(n) * 4
    return verts, polys, pv.reshape(-1, 6)


def gen_c2(n_tris: int = 1_000_000, width: int = 1920, height: int = 1080, log_focus: int = 10,
           radius: int = 8, seed: int = 0xC0FFEE02, cull: str = "random") -> Scene:
    """Gouraud-shaded Z-buffered triangles: centres uniform over the screen plus a
    32-pixel margin, depth 64..4095, vertices within +-radius pixels of the centre,
    per-vertex RGB uniform in 0..255, white base colour, identity camera."""
    rng = Rng(seed)
    verts, polys, pv = _triangles(rng, n_tris, width, height, log_focus, radius,
                                  (-32, width + 31), (-32, height + 31), (64, 4095), cull)
    return Scene(width, height, log_focus, M_SHADED, verts, polys, pv, matrix=IDENTITY.copy(),
                 name=f"c2-r{radius}-{n_tris}")


def gen_c3(n_tris: int = 10_000_000, width: int = 3840, height: int = 2160, log_focus: int = 11,
           radius: int = 8, seed: int = 0xC0FFEE03, equal_lights: bool = True,
           lin_percent: int = 90, cull: str = "random", regular_mapping: bool = False) -> Scene:
    """Texture-mapped triangles that exercise the view-volume, near-plane and screen
    clippers: centres uniform over the screen plus a quarter-screen margin, depth
    4..4095, one 256x256 texture, texture coordinates 0..1023.  `lin_percent` of the
    triangles keep their first vertex at z >= 1024 so that they take the linear mapper.
    equal_lights=True gives each triangle one light value for all three vertices.
    regular_mapping=True keeps out the (few per million) random texture mappings on which
    the reference itself divides by zero."""
    rng = Rng(seed)
    verts, polys, pv = _triangles(rng, n_tris, width, height, log_focus, radius,
                                  (-(width // 4), width + width // 4 - 1),
                                  (-(height // 4), height + height // 4 - 1), (4, 4095), cull,
                                  rgb_equal=equal_lights, tex=True, first_z_min=1024,
                                  first_z_frac=lin_percent, regular_mapping=regular_mapping)
    aux = np.zeros((n_tris, 4), dtype=np.int32)
    aux[:, 3] = M_PLANNAR | (8 << 8)
    return Scene(width, height, log_focus, M_TEXTURED, verts, polys, pv, polyaux=aux,
                 textures=[(make_texture(seed ^ 0x7E47, 8), 8)], matrix=IDENTITY.copy(),
                 name=f"c3-r{radius}-{n_tris}")


def gen_fuzz_polys(n_polys: int, width: int, height: int, log_focus: int, seed: int,
                   render_type: int = M_SHADED, max_edges: int = 6, textured: bool = False,
                   equal_lights: bool = False, zrange=(12, 900)) -> Scene:
    """Differential-test scene: convex-ish and irregular polygons of 3..max_edges
    vertices, many straddling the screen border or the near plane, some degenerate
    (repeated vertices, zero area, zero height), with shared vertices between
    neighbouring polygons so that depth and ownership ties occur."""
    rng = Rng(seed)
    verts_l, polys_l, pv_l, aux_l = [], [], [], []
    first = 0
    nv = 0
    prev_pts = None
    for i in range(n_polys):
        e = int(rng.ints(3, max_edges, 1)[0])
        mode = int(rng.ints(0, 9, 1)[0])
        zc = int(rng.ints(zrange[0], zrange[1], 1)[0]) if mode != 3 else int(rng.ints(2, 30, 1)[0])
        rad = int(rng.ints(1, max(2, min(width, height) // 2), 1)[0])
        if mode == 4:
            rad = int(rng.ints(0, 3, 1)[0])
        cxp = int(rng.ints(-width // 3, width + width // 3, 1)[0])
        cyp = int(rng.ints(-height // 3, height + height // 3, 1)[0])
        ang0 = rng.ints(0, 359, 1)[0]
        pts = []
        for k in range(e):
            a = (ang0 + (360 * k) // e + int(rng.ints(-20, 20, 1)[0])) * np.pi / 180.0
            rr = rad if mode != 5 else int(rng.ints(0, rad, 1)[0])
            # clockwise on screen (y down) gives a negative cross product = drawn
            px = cxp + int(round(rr * np.cos(a)))
            py = cyp + int(round(rr * np.sin(a)))
            zv = zc + int(rng.ints(-8, 8, 1)[0]) if mode != 3 else zc + int(rng.ints(-25, 25, 1)[0])
            if mode == 6 and k > 0:
                py = cyp                                            # zero height
            pts.append((px, py, zv))
        if mode == 7 and e > 3:
            pts[2] = pts[1]                                         # repeated vertex
        if mode == 8 and prev_pts is not None:
            # share an edge (two screen points, same depth) with the previous polygon
            pts[0], pts[1] = prev_pts[1], prev_pts[0]
        prev_pts = list(pts)
        if int(rng.ints(0, 3, 1)[0]) == 0:
            pts = pts[::-1]                                         # back-facing
        idxs = []
        for (px, py, zv) in pts:
            vx = ((px - width // 2) * zv) >> log_focus
            vy = ((py - height // 2) * zv) >> log_focus
            verts_l.append((vx, vy, zv, 0))
            idxs.append(nv)
            nv += 1
        lights = rng.ints(0, 300, 3 * e).reshape(e, 3)
        if equal_lights:
            lights[:] = lights[0]
        tx = rng.ints(-64, 1200, 2 * e).reshape(e, 2)
        for k in range(e + 1):
            kk = k % e
            pv_l.append((idxs[kk], lights[kk, 0], lights[kk, 1], lights[kk, 2], tx[kk, 0], tx[kk, 1]))
        col = int(rng.ints(0, 0xFFFFFF, 1)[0])
        polys_l.append((col, 0 if textured else -1, e, first))
        aux_l.append((int(rng.ints(0, 255, 1)[0]), int(rng.ints(0, 255, 1)[0]),
                      int(rng.ints(0, 255, 1)[0]), M_PLANNAR | (8 << 8)))
        first += e + 1
    sc = Scene(width, height, log_focus, render_type,
               np.array(verts_l, dtype=np.int32).reshape(-1, 4),
               np.array(polys_l, dtype=np.int32).reshape(-1, 4),
               np.array(pv_l, dtype=np.int32).reshape(-1, 6),
               polyaux=np.array(aux_l, dtype=np.int32).reshape(-1, 4),
               matrix=IDENTITY.copy(), name=f"fuzz-{seed:x}")
    if textured:
        sc.textures = [(make_texture(seed ^ 0x7E47, 6), 6)]
    return sc


def gen_screen_polys(n: int, width: int, height: int, seed: int, dim: int = 6, max_edges: int = 6,
                     spread: int = 2):
    """Already-projected closed polygons for the G_*_polygon level: a list of
    (length, tuples int32[length+1][dim], colour).  Coordinates reach `spread` screen
    sizes beyond the border so that both 2-D clippers run; depths cover the range the
    perspective stage produces (-52428..0) and a little beyond."""
    rng = Rng(seed)
    out = []
    for _ in range(n):
        e = int(rng.ints(3, max_edges, 1)[0])
        big = int(rng.ints(0, 3, 1)[0])
        if big == 0:
            cx = int(rng.ints(-width * spread // 2, width + width * spread // 2, 1)[0])
            cy = int(rng.ints(-height * spread // 2, height + height * spread // 2, 1)[0])
            rad = int(rng.ints(1, max(width, height), 1)[0])
        else:
            cx = int(rng.ints(-8, width + 8, 1)[0])
            cy = int(rng.ints(-8, height + 8, 1)[0])
            rad = int(rng.ints(0, 24, 1)[0])
        t = np.zeros((e + 1, dim), dtype=np.int32)
        for k in range(e):
            t[k, 0] = cx + int(rng.ints(-rad, rad, 1)[0])
            t[k, 1] = cy + int(rng.ints(-rad, rad, 1)[0])
            t[k, 2] = int(rng.ints(-53000, 100, 1)[0])
            for j in range(3, min(dim, 6)):
                t[k, j] = int(rng.ints(0, 300, 1)[0]) << 16
            for j in range(6, dim):
                t[k, j] = int(rng.ints(-64, 1200, 1)[0]) << 16
        t[e] = t[0]
        out.append((e, t, int(rng.ints(0, 0xFFFFFF, 1)[0])))
    return out


def frame_digest(a: np.ndarray) -> str:
    """Checksum of an int32 frame (little-endian bytes): blake2b-64, hex."""
    import hashlib
    return hashlib.blake2b(np.ascontiguousarray(a, dtype="<i4").tobytes(), digest_size=8).hexdigest()


def gen_shading_case(n_polys: int, seed: int, n_verts: int = 0, n_normals: int = 0, extent: int = 2000,
                     far_light: bool = True):
    """Input of the shading pre-pass (M_shade_polygon_object): random polygons of 3..5 edges
    over shared vertices, unit normals scaled by 1<<7, one ambient, two point and one
    directional light (a far point light exercises the long-vector branch of T_norm).
    -> dict(verts, normals, polys, polyverts, polynormals, lights)"""
    rng = Rng(seed)
    nv = n_verts or max(8, n_polys * 2)
    nn = n_normals or max(8, n_polys)
    verts = np.zeros((nv, 4), np.int32)
    verts[:, :3] = rng.ints(-extent, extent, nv * 3).reshape(-1, 3)
    raw = rng.ints(-100, 100, nn * 3).reshape(-1, 3).astype(np.float64) + 0.5
    normals = np.zeros((nn, 4), np.int32)
    normals[:, :3] = np.round(raw / np.linalg.norm(raw, axis=1)[:, None] * 128).astype(np.int32)
    e = rng.ints(3, 5, n_polys).astype(np.int64)
    first = np.concatenate([[0], np.cumsum(e + 1)[:-1]])
    npv = int((e + 1).sum())
    pv = np.zeros((npv, 6), np.int32)
    pv[:, 0] = rng.ints(0, nv - 1, npv)
    pv[:, 1:4] = -1                                   # overwritten by the pre-pass
    pv[:, 4:6] = rng.ints(0, 1023, npv * 2).reshape(-1, 2)
    pn = rng.ints(0, nn - 1, npv).astype(np.int32)
    polys = np.zeros((n_polys, 4), np.int32)
    polys[:, 0] = 0x00FFFFFF
    polys[:, 1] = -1
    polys[:, 2] = e
    polys[:, 3] = first
    lights = np.array([[1, 0, 0, 0, 0, 20, 30, 40],
                       [2, 500, -300, -800, 0, 200, 180, 160],
                       [4, 40, -90, 80, 0, 90, 100, 110],
                       [2, 70000 if far_light else 900, 3, 9, 0, 255, 255, 255]], np.int32)
    return dict(verts=verts, normals=normals, polys=polys, polyverts=pv, polynormals=pn, lights=lights)


def gen_sphere_object(n_long: int = 16, n_lat: int = 8, radius: int = 300, centre=(0, 0, 0), with_quad: bool = True):
    """A closed lat/long sphere of M_CURVED triangles over shared vertices (two poles and
    n_lat-1 rings of n_long vertices) and, optionally, one M_PLANNAR quad below it: the kind
    of object M_init_polygon_object gives normals to.
    -> dict(verts int32[nv][4], polys int32[np][4], types int32[np], polyverts int32[*][6])"""
    cx, cy, cz = centre
    vs = [(cx, cy - radius, cz)]
    for i in range(1, n_lat):
        th = np.pi * i / n_lat
        for j in range(n_long):
            ph = 2 * np.pi * j / n_long
            vs.append((cx + int(round(radius * np.sin(th) * np.cos(ph))), cy - int(round(radius * np.cos(th))),
                       cz + int(round(radius * np.sin(th) * np.sin(ph)))))
    vs.append((cx, cy + radius, cz))
    south = len(vs) - 1

    def ring(i, j):
        return 1 + (i - 1) * n_long + (j % n_long)

    faces, types = [], []
    for j in range(n_long):
        faces.append((0, ring(1, j + 1), ring(1, j)))
        faces.append((south, ring(n_lat - 1, j), ring(n_lat - 1, j + 1)))
        for i in range(1, n_lat - 1):
            faces.append((ring(i, j), ring(i, j + 1), ring(i + 1, j + 1)))
            faces.append((ring(i, j), ring(i + 1, j + 1), ring(i + 1, j)))
    types += [2] * len(faces)
    if with_quad:
        b = len(vs)
        y = cy + radius + 30
        vs += [(cx - 2 * radius, y, cz + 2 * radius), (cx + 2 * radius, y, cz + 2 * radius),
               (cx + 2 * radius, y, cz - 2 * radius), (cx - 2 * radius, y, cz - 2 * radius)]
        faces.append((b, b + 1, b + 2, b + 3))
        types.append(1)
    verts = np.zeros((len(vs), 4), np.int32)
    verts[:, :3] = np.array(vs, np.int32)
    recs, polys = [], []
    for f in faces:
        polys.append((0x00FFFFFF, -1, len(f), len(recs)))
        for idx in list(f) + [f[0]]:
            recs.append((idx, 0, 0, 0, 0, 0))
    return dict(verts=verts, polys=np.array(polys, np.int32), types=np.array(types, np.int32),
                polyverts=np.array(recs, np.int32))


def gen_bicubic_patch(seed: int, extent: int = 300, bulge: int = 150, centre=(0, 0, 900)):
    """16 control points int32[16][4] of a bicubic Bezier patch: a 4x4 grid in x, y around
    `centre` with random displacements in all three coordinates."""
    rng = Rng(seed)
    c = np.zeros((16, 4), np.int32)
    for i in range(4):
        for j in range(4):
            c[i * 4 + j, 0] = centre[0] + (j - 1.5) * 2 * extent / 3 + int(rng.ints(-bulge // 4, bulge // 4, 1)[0])
            c[i * 4 + j, 1] = centre[1] + (i - 1.5) * 2 * extent / 3 + int(rng.ints(-bulge // 4, bulge // 4, 1)[0])
            c[i * 4 + j, 2] = centre[2] + int(rng.ints(-bulge, bulge, 1)[0])
    return c


def tracer_polygon_objects(lights, spheres, quad, pcol):
    """The ray tracer's world of tracer.dat as polygon objects (configuration 1): the quad and
    three lat/long spheres (32 x 16 triangles), vertex light from the point lights (Lambert,
    ambient 0.1), for M_SHADED at 640x480.  lights: (x, y, z, r, g, b) each; spheres:
    (r, g, b, cx, cy, cz, radius) each; quad: float[5][3], the closed vertex list of the
    file; pcol: the polygon's colour.  The harness defines this tessellation (the reference
    only ray-traces that world); golden frames pin it."""
    objs = []

    def shade(p, n):
        tot = 0.1
        for (lx, ly, lz, ir, ig, ib) in lights:
            d = np.array([lx, ly, lz]) - p
            d = d / np.linalg.norm(d)
            tot += max(0.0, float(np.dot(d, n))) * ir
        return int(min(255, max(0, round(255 * tot))))

    # quad (closed list of 5 vertices in the file; 4 edges)
    qn = np.array([0.0, -1.0, 0.0])
    verts = [(int(x), int(y), int(z), 0) for (x, y, z) in quad[:4]]
    recs = []
    for k in [0, 1, 2, 3, 0]:
        l = shade(quad[k], qn)
        recs.append((k, l, l, l, 0, 0))
    colour = (int(pcol[0] * 255) << 16) | (int(pcol[1] * 255) << 8) | int(pcol[2] * 255)
    # both windings so that one of them faces the camera
    polys = [(colour, -1, 4, 0), (colour, -1, 4, 5)]
    recs2 = [recs[0], recs[3], recs[2], recs[1], recs[0]]
    objs.append(dict(verts=np.array(verts, np.int32), polys=np.array(polys, np.int32),
                     polyverts=np.array(recs + recs2, np.int32)))
    # spheres
    NLON, NLAT = 32, 16
    for (cr, cg, cb, cx, cy, cz, rad) in spheres:
        colour = (min(255, int(cr * 255)) << 16) | (min(255, int(cg * 255)) << 8) | min(255, int(cb * 255))
        vs, light = [], []

        def add(theta, phi):
            n = np.array([np.sin(theta) * np.cos(phi), np.cos(theta), np.sin(theta) * np.sin(phi)])
            p = np.array([cx, cy, cz]) + rad * n
            vs.append((int(round(p[0])), int(round(p[1])), int(round(p[2])), 0))
            light.append(shade(p, n))
            return len(vs) - 1

        top = add(0.0, 0.0)
        rings = [[add(np.pi * i / NLAT, 2 * np.pi * j / NLON) for j in range(NLON)] for i in range(1, NLAT)]
        bot = add(np.pi, 0.0)
        tris = []
        for j in range(NLON):
            j2 = (j + 1) % NLON
            tris.append((top, rings[0][j2], rings[0][j]))
            tris.append((bot, rings[-1][j], rings[-1][j2]))
            for i in range(NLAT - 2):
                a, b, c, d = rings[i][j], rings[i][j2], rings[i + 1][j], rings[i + 1][j2]
                tris.append((a, b, d))
                tris.append((a, d, c))
        recs, polys = [], []
        for (a, b, c) in tris:
            polys.append((colour, -1, 3, len(recs)))
            for k in (a, b, c, a):
                recs.append((k, light[k], light[k], light[k], 0, 0))
        objs.append(dict(verts=np.array(vs, np.int32), polys=np.array(polys, np.int32),
                         polyverts=np.array(recs, np.int32)))
    return objs
