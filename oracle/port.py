"""ctypes wrapper around oracle/lt_oracle.c, the CPU restatement used as the checker."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblt_oracle.so")
_SRC = os.path.join(_HERE, "lt_oracle.c")

I32P = C.POINTER(C.c_int32)


def build(force=False):
    """Compile the restatement (gcc is on every box this runs on)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", "-O2", "-fwrapv", "-shared", "-fPIC", _SRC, "-o", _SO, "-lm"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.lto_create.restype = C.c_void_p
        L.lto_create.argtypes = [C.c_int, C.c_int]
        L.lto_destroy.argtypes = [C.c_void_p]
        L.lto_cbuf.restype = I32P
        L.lto_cbuf.argtypes = [C.c_void_p]
        L.lto_zbuf.restype = I32P
        L.lto_zbuf.argtypes = [C.c_void_p]
        L.lto_set_fast_reset.argtypes = [C.c_void_p, C.c_int]
        L.lto_set_texture.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.lto_counters.argtypes = [C.c_void_p, C.c_void_p]
        L.lto_reset_counters.argtypes = [C.c_void_p]
        L.lto_clear.argtypes = [C.c_void_p] + [C.c_int32] * 4
        L.lto_cl_light.restype = C.c_int32
        L.lto_cl_light.argtypes = [C.c_int32] * 4
        L.lto_cl_colour.restype = C.c_int32
        L.lto_cl_colour.argtypes = [C.c_int32] * 3
        L.lto_trig_tables.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.lto_make_matrix.argtypes = [C.c_int] + [C.c_int] * 3 + [C.c_int32] * 3 + [C.c_int] * 3 + [C.c_int32] * 3 + [C.c_void_p]
        L.lto_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.lto_perspective.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.lto_normal_z_negative.restype = C.c_int
        L.lto_normal_z_negative.argtypes = [C.c_void_p] * 3
        L.lto_volume_clipping.restype = C.c_int
        L.lto_volume_clipping.argtypes = [C.c_void_p] * 2
        L.lto_line_clipping.restype = C.c_int
        L.lto_line_clipping.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.lto_polygon_z_clipping.restype = C.c_int
        L.lto_polygon_z_clipping.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.lto_polygon_x_clipping.restype = C.c_int
        L.lto_polygon_x_clipping.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.lto_scan_polygon.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6
        L.lto_g_polygon.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_int32] * 4 + [C.c_int]
        L.lto_g_prp_polygon.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.lto_render_polys.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.c_int] * 4
        L.lto_self_matrix.argtypes = [C.c_int] * 4 + [C.c_void_p]
        L.lto_shade_polys.restype = C.c_int
        L.lto_shade_polys.argtypes = ([C.c_void_p] + [C.c_int32] * 3 + [C.c_void_p, C.c_int] * 2 + [C.c_void_p, C.c_int]
                                      + [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p])
        L.lto_init_bicubic.restype = C.c_int
        L.lto_init_bicubic.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.lto_shade_bicubic.restype = C.c_int
        L.lto_shade_bicubic.argtypes = [C.c_void_p] + [C.c_int32] * 3 + [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.lto_render_bicubic.restype = C.c_int
        L.lto_render_bicubic.argtypes = ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int32, C.c_int, C.c_int, C.c_void_p]
                                         + [C.c_int] * 3)
        L.lto_set_clip_window.restype = None
        L.lto_set_clip_window.argtypes = [C.c_void_p] + [C.c_int32] * 4
        L.lto_g_dot.restype = None
        L.lto_g_dot.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int32] * 4
        L.lto_g_pixel.restype = None
        L.lto_g_pixel.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.lto_g_line.restype = None
        L.lto_g_line.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int32] * 4
        L.lto_normal_plane.restype = C.c_int
        L.lto_normal_plane.argtypes = [C.c_void_p] * 4
        L.lto_init_normals.restype = C.c_int
        L.lto_init_normals.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


WORK_TUPLES = 64


class Oracle:
    """One framebuffer of the CPU restatement (run-time width x height)."""

    def __init__(self, w, h):
        self.L = lib()
        self.w, self.h = w, h
        self.c = self.L.lto_create(w, h)
        self._keep = {}

    def close(self):
        if self.c:
            self.L.lto_destroy(self.c)
            self.c = None

    def __del__(self):
        self.close()

    # --- framebuffer
    def clear(self, colour, r, g, b):
        self.L.lto_clear(self.c, colour, r, g, b)

    def cbuf(self):
        return np.ctypeslib.as_array(self.L.lto_cbuf(self.c), shape=(self.h, self.w))

    def zbuf(self):
        return np.ctypeslib.as_array(self.L.lto_zbuf(self.c), shape=(self.h, self.w))

    def counters(self):
        out = np.zeros(6, dtype=np.uint64)
        self.L.lto_counters(self.c, _p(out))
        return dict(zip(["fragments", "passes", "polys_in", "polys_drawn", "overflow", "unsupported"], map(int, out)))

    def reset_counters(self):
        self.L.lto_reset_counters(self.c)

    def set_fast_reset(self, on):
        self.L.lto_set_fast_reset(self.c, int(on))

    def set_texture(self, handle, texels, log_size):
        t = _i32(texels).reshape(-1)
        assert t.size == 1 << (2 * log_size)
        self._keep[handle] = t
        self.L.lto_set_texture(self.c, handle, _p(t), log_size)

    # --- stages
    def perspective(self, tuples, dim, log_focus):
        t = _i32(tuples).reshape(-1, dim)
        out = np.zeros_like(t)
        self.L.lto_perspective(self.c, _p(t), _p(out), dim, t.shape[0], log_focus)
        return out

    def line_clipping(self, axis, v1, v2):
        v1, v2 = _i32(v1), _i32(v2)
        dim = v1.size
        o1, o2 = np.zeros(dim, np.int32), np.zeros(dim, np.int32)
        moved = C.c_int(0)
        r = self.L.lto_line_clipping(self.c, axis, _p(v1), _p(v2), dim, _p(o1), _p(o2), C.byref(moved))
        return r, o1, o2, moved.value

    def polygon_clipping(self, axis, tuples, dim):
        """tuples: (length+1, dim) closed polygon -> (n, clipped (n+1, dim))."""
        t = _i32(tuples).reshape(-1, dim)
        out = np.zeros(((WORK_TUPLES + 1), dim), np.int32)
        f = self.L.lto_polygon_z_clipping if axis == 2 else self.L.lto_polygon_x_clipping
        n = f(self.c, _p(t), _p(out), dim, t.shape[0] - 1)
        return n, out[: n + 1].copy()

    def scan_polygon(self, tuples, dim):
        t = _i32(tuples).reshape(-1, dim)
        xs = np.zeros(self.h, np.int32)
        xe = np.zeros(self.h, np.int32)
        vs = np.zeros((self.h, 6), np.int32)
        ve = np.zeros((self.h, 6), np.int32)
        mn, mx = C.c_int(0), C.c_int(0)
        self.L.lto_scan_polygon(self.c, _p(t), t.shape[0] - 1, dim, C.byref(mn), C.byref(mx), _p(xs), _p(xe), _p(vs), _p(ve))
        return mn.value, mx.value, xs, xe, vs, ve

    # --- draws
    def g_polygon(self, kind, tuples, colour=0, r=0, g=0, b=0, tex=0):
        dim = {2: 3, 4: 6, 8: 8}[kind]
        t = _i32(tuples).reshape(-1, dim)
        self.L.lto_g_polygon(self.c, kind, _p(t), t.shape[0] - 1, colour, r, g, b, tex)

    def g_prp_polygon(self, tuples, O, u, v, tex, log_scale, log_focus):
        t = _i32(tuples).reshape(-1, 8)
        O, u, v = _i32(O), _i32(u), _i32(v)
        self.L.lto_g_prp_polygon(self.c, _p(t), t.shape[0] - 1, _p(O), _p(u), _p(v), tex, log_scale, log_focus)

    def set_clip_window(self, minx, miny, maxx, maxy):
        """C_init_clipping"""
        self.L.lto_set_clip_window(self.c, int(minx), int(miny), int(maxx), int(maxy))

    def g_line(self, v1, v2, colour, r=0, g=0, b=0):
        """G_line on projected {X, Y, Z} end points"""
        a, bb = _i32(v1), _i32(v2)
        self.L.lto_g_line(self.c, _p(a), _p(bb), int(colour), int(r), int(g), int(b))

    def bicubic(self, controls, log_render_size, colour, log_texture_space_size, tex, pos, ang, lights, render_type,
                log_focus, force_linear=1, variant=0):
        """M_init_bicubic + M_shade_bicubic + M_render_bicubic of one patch into this frame ->
        (normals int32[size*size][4], intensities int32[size*size][3])"""
        c, li = _i32(controls).reshape(16, 4), _i32(lights).reshape(-1, 8)
        size = (1 << log_render_size) + 1
        normals = np.zeros((size * size, 4), np.int32)
        inten = np.zeros((size * size, 3), np.int32)
        if self.L.lto_init_bicubic(_p(c), log_render_size, _p(normals)) != 0:
            raise ValueError("M_init_bicubic stops on this patch")
        s9 = self_matrix(*ang, variant=variant)
        if self.L.lto_shade_bicubic(_p(s9), int(pos[0]), int(pos[1]), int(pos[2]), _p(c), log_render_size, _p(normals),
                                    _p(li), li.shape[0], _p(inten)) != 0:
            raise ValueError("M_shade_bicubic stops on this patch")
        m = make_matrix((0, 0, 0, 0, 0, 0), (*ang, *pos), variant)
        if self.L.lto_render_bicubic(self.c, _p(m), _p(c), log_render_size, int(colour), log_texture_space_size, int(tex),
                                     _p(inten), int(render_type), int(log_focus), int(force_linear)) != 0:
            raise ValueError("M_render_bicubic stops on this patch")
        return normals, inten

    def g_dot(self, v, colour, r=0, g=0, b=0):
        a = _i32(v)
        self.L.lto_g_dot(self.c, _p(a), int(colour), int(r), int(g), int(b))

    def g_pixel(self, v, colour):
        a = _i32(v)
        self.L.lto_g_pixel(self.c, _p(a), int(colour))

    def render_polys(self, viewverts, polys, polyaux, polyverts, render_type, log_focus, force_linear=0):
        vv, po, pv = _i32(viewverts), _i32(polys), _i32(polyverts)
        pa = None if polyaux is None else _i32(polyaux)
        self.L.lto_render_polys(self.c, _p(vv), _p(po), _p(pa), _p(pv), po.shape[0], render_type, log_focus, force_linear)


def cl_light(colour, r, g, b):
    return lib().lto_cl_light(colour, r, g, b)


def cl_colour(r, g, b):
    return lib().lto_cl_colour(r, g, b)


def trig_tables(variant=0):
    s, c = np.zeros(256, np.int32), np.zeros(256, np.int32)
    lib().lto_trig_tables(variant, _p(s), _p(c))
    return s, c


def make_matrix(cam=(0, 0, 0, 0, 0, 0), obj=(0, 0, 0, 0, 0, 0), variant=0):
    """cam = (alp, bet, gam, x, y, z); obj = (alp, bet, gam, x, y, z) -> int32[12]."""
    m = np.zeros(12, np.int32)
    lib().lto_make_matrix(variant, *[int(v) for v in cam],
                          *[int(v) for v in obj], _p(m))
    return m


def transform(m12, verts):
    v = _i32(verts).reshape(-1, 4)
    out = v.copy()
    m = _i32(m12)
    lib().lto_transform(_p(m), _p(v), _p(out), v.shape[0])
    return out


def normal_z_negative(a, b, c):
    a, b, c = _i32(a), _i32(b), _i32(c)
    return lib().lto_normal_z_negative(_p(a), _p(b), _p(c))


def volume_clipping(mn, mx):
    mn, mx = _i32(mn), _i32(mx)
    return lib().lto_volume_clipping(_p(mn), _p(mx))


def self_matrix(alp, bet, gam, variant=0):
    """T_set_self_rotation -> int32[9] = sx1 sy1 sz1 sx2 sy2 sz2 sx3 sy3 sz3."""
    s = np.zeros(9, np.int32)
    lib().lto_self_matrix(variant, int(alp), int(bet), int(gam), _p(s))
    return s


def shade_polys(s9, pos, verts, normals, polys, polyverts, polynormals, polyaux, lights):
    """M_shade_polygon_object -> (polyverts with R,G,B, polyaux with flat r,g,b, skipped, bad_polys).
    lights: int32[nl][8] = {type, px, py, pz, pw, red, green, blue}."""
    s9, v, n, po = _i32(s9), _i32(verts).reshape(-1, 4), _i32(normals).reshape(-1, 4), _i32(polys).reshape(-1, 4)
    pv, pn = _i32(polyverts).reshape(-1, 6).copy(), _i32(polynormals).reshape(-1)
    pa = np.zeros((po.shape[0], 4), np.int32) if polyaux is None else _i32(polyaux).reshape(-1, 4).copy()
    li = _i32(lights).reshape(-1, 8)
    bad = C.c_int(0)
    skipped = lib().lto_shade_polys(_p(s9), int(pos[0]), int(pos[1]), int(pos[2]), _p(v), v.shape[0], _p(n), n.shape[0],
                                    _p(po), po.shape[0], _p(pv), _p(pn), pv.shape[0], _p(pa), _p(li), li.shape[0],
                                    C.byref(bad))
    return pv, pa, skipped, bad.value


def normal_plane(p1, p2, p3):
    """T_normal_plane -> int32[3], or None for a zero-length normal."""
    a, b, c, to = _i32(p1), _i32(p2), _i32(p3), np.zeros(3, np.int32)
    return to if lib().lto_normal_plane(_p(a), _p(b), _p(c), _p(to)) else None


def init_normals(verts, polys, types, polyverts, limit=1024):
    """The normals half of M_init_polygon_object -> (normals int32[nn][3], polynormals
    int32[records], -2 where a polygon got no m_normals array).  Raises ValueError where the
    reference stops (zero-length plane normal, list full)."""
    v, po, ty = _i32(verts).reshape(-1, 4), _i32(polys).reshape(-1, 4), _i32(types).reshape(-1)
    pv = _i32(polyverts).reshape(-1, 6)
    normals = np.zeros((limit, 4), np.int32)
    pn = np.full(pv.shape[0], -2, np.int32)
    nn = lib().lto_init_normals(_p(v), v.shape[0], _p(po), _p(ty), po.shape[0], _p(pv), _p(normals), limit, _p(pn))
    if nn < 0:
        raise ValueError("zero-length plane normal" if nn == -1 else "normal list full")
    return normals[:nn, :3].copy(), pn

