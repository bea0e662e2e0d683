"""ctypes wrapper around the checker libraries built from the reference's own sources.

oracle/build_ref.sh compiles the reference where it lies under /root/reference into
oracle/_ref/libltref_<W>x<H>_<tables>.so (one per screen size: the reference fixes it
at compile time).  The libraries travel with the tree; the sources never do.  Where
neither a prebuilt library nor the reference tree is present, `available()` is False
and callers skip.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REFDIR = os.environ.get("LT_REFERENCE_DIR", "/root/reference")

I32P = C.POINTER(C.c_int32)


def so_path(w, h, tables="trans"):
    return os.path.join(_HERE, "_ref", f"libltref_{w}x{h}_{tables}.so")


def available(w, h, tables="trans"):
    return os.path.exists(so_path(w, h, tables)) or os.path.isdir(_REFDIR)


def build(w, h, tables="trans"):
    p = so_path(w, h, tables)
    if not os.path.exists(p):
        subprocess.check_call([os.path.join(_HERE, "build_ref.sh"), str(w), str(h), tables])
    return p


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


_libs = {}


class Ref:
    """The reference's own code at one compile-time screen size.  Its state is global
    to the loaded library (as in the reference), so there is one instance per size."""

    def __new__(cls, w, h, tables="trans"):
        key = (w, h, tables)
        if key in _libs:
            return _libs[key]
        self = super().__new__(cls)
        L = C.CDLL(build(w, h, tables))
        L.ltref_cbuf.restype = I32P
        L.ltref_zbuf.restype = I32P
        for f in ("ltref_cl_light", "ltref_cl_colour", "ltref_volume_clipping", "ltref_polygon_z_clipping",
                  "ltref_polygon_x_clipping", "ltref_normal_z_negative", "ltref_line_clipping"):
            getattr(L, f).restype = C.c_int
        L.ltref_init()
        assert L.ltref_width() == w and L.ltref_height() == h
        self.L, self.w, self.h = L, w, h
        _libs[key] = self
        return self

    def clear(self, colour, r, g, b):
        self.L.ltref_clear(C.c_int(colour), r, g, b)

    def cbuf(self):
        return np.ctypeslib.as_array(self.L.ltref_cbuf(), shape=(self.h, self.w))

    def zbuf(self):
        return np.ctypeslib.as_array(self.L.ltref_zbuf(), shape=(self.h, self.w))

    def cl_light(self, colour, r, g, b):
        return self.L.ltref_cl_light(C.c_int(colour), r, g, b)

    def cl_colour(self, r, g, b):
        return self.L.ltref_cl_colour(r, g, b)

    def trig_tables(self):
        s, c = np.zeros(256, np.int32), np.zeros(256, np.int32)
        self.L.ltref_trig_tables(_p(s), _p(c))
        return s, c

    def set_texture(self, handle, texels, log_size):
        t = _i32(texels).reshape(-1)
        self.L.ltref_set_texture(handle, _p(t), log_size)

    def make_matrix(self, cam=(0, 0, 0, 0, 0, 0), obj=(0, 0, 0, 0, 0, 0), log_focus=8):
        m = np.zeros(12, np.int32)
        self.L.ltref_make_matrix(*[int(v) for v in cam], log_focus,
                                 *[int(v) for v in obj], _p(m))
        return m

    def transform(self, m12, verts):
        v = _i32(verts).reshape(-1, 4).copy()
        out = v.copy()
        m = _i32(m12)
        self.L.ltref_set_matrix(_p(m))
        self.L.ltref_transform(_p(v), _p(out), v.shape[0])
        return out

    def perspective(self, tuples, dim, log_focus):
        t = _i32(tuples).reshape(-1, dim).copy()
        out = np.zeros_like(t)
        self.L.ltref_perspective(_p(t), _p(out), dim, t.shape[0], log_focus)
        return out

    def normal_z_negative(self, a, b, c):
        a, b, c = _i32(a).copy(), _i32(b).copy(), _i32(c).copy()
        return self.L.ltref_normal_z_negative(_p(a), _p(b), _p(c))

    def volume_clipping(self, mn, mx):
        mn, mx = _i32(mn).copy(), _i32(mx).copy()
        return self.L.ltref_volume_clipping(_p(mn), _p(mx))

    def line_clipping(self, axis, v1, v2):
        v1, v2 = _i32(v1).copy(), _i32(v2).copy()
        dim = v1.size
        o1, o2 = np.zeros(dim, np.int32), np.zeros(dim, np.int32)
        moved = C.c_int(0)
        r = self.L.ltref_line_clipping(axis, _p(v1), _p(v2), dim, _p(o1), _p(o2), C.byref(moved))
        return r, o1, o2, moved.value

    def polygon_clipping(self, axis, tuples, dim):
        t = _i32(tuples).reshape(-1, dim).copy()
        out = np.zeros((65, dim), np.int32)     # the reference's own arrays hold 16
        f = self.L.ltref_polygon_z_clipping if axis == 2 else self.L.ltref_polygon_x_clipping
        n = f(_p(t), _p(out), dim, t.shape[0] - 1)
        return n, out[: n + 1].copy()

    def scan_polygon(self, tuples, dim):
        t = _i32(tuples).reshape(-1, dim).copy()
        xs = np.zeros(self.h, np.int32)
        xe = np.zeros(self.h, np.int32)
        vs = np.zeros((self.h, 6), np.int32)
        ve = np.zeros((self.h, 6), np.int32)
        mn, mx = C.c_int(0), C.c_int(0)
        self.L.ltref_scan_polygon(_p(t), t.shape[0] - 1, dim, C.byref(mn), C.byref(mx), _p(xs), _p(xe), _p(vs), _p(ve))
        return mn.value, mx.value, xs, xe, vs, ve

    def g_polygon(self, kind, tuples, colour=0, r=0, g=0, b=0, tex=0):
        dim = {2: 3, 4: 6, 8: 8}[kind]
        t = _i32(tuples).reshape(-1, dim).copy()
        self.L.ltref_g_polygon(kind, _p(t), t.shape[0] - 1, C.c_int(colour), r, g, b, tex)

    def g_prp_polygon(self, tuples, O, u, v, tex, log_scale, log_focus):
        t = _i32(tuples).reshape(-1, 8).copy()
        O, u, v = _i32(O).copy(), _i32(u).copy(), _i32(v).copy()
        self.L.ltref_g_prp_polygon(_p(t), t.shape[0] - 1, _p(O), _p(u), _p(v), tex, log_scale, log_focus)

    def render_polys(self, viewverts, polys, polyaux, polyverts, render_type, log_focus, force_linear=0):
        vv, po, pv = _i32(viewverts).copy(), _i32(polys), _i32(polyverts).copy()
        pa = None if polyaux is None else _i32(polyaux)
        self.L.ltref_render_polys(_p(vv), _p(po), _p(pa), _p(pv), po.shape[0], render_type, log_focus, force_linear)

    def render_object(self, verts, polys, polyaux, polyverts, render_type, pos, ang):
        v, po, pv = _i32(verts).copy(), _i32(polys), _i32(polyverts).copy()
        pa = None if polyaux is None else _i32(polyaux)
        self.L.ltref_render_object(_p(v), v.shape[0], _p(po), _p(pa), _p(pv), po.shape[0], render_type,
                                   pos[0], pos[1], pos[2], ang[0], ang[1], ang[2])

    def self_matrix(self, alp, bet, gam):
        s = np.zeros(9, np.int32)
        self.L.ltref_self_matrix(int(alp), int(bet), int(gam), _p(s))
        return s

    def shade_object(self, verts, normals, polys, polyverts, polynormals, polyaux, pos, ang, lights):
        """the real M_shade_polygon_object (at most 1024 vertices and normals) ->
        (polyverts with R,G,B, polyaux with flat r,g,b)"""
        v, n, po = _i32(verts).reshape(-1, 4).copy(), _i32(normals).reshape(-1, 4).copy(), _i32(polys).reshape(-1, 4)
        assert v.shape[0] <= 1024 and n.shape[0] <= 1024
        pv, pn = _i32(polyverts).reshape(-1, 6).copy(), _i32(polynormals).reshape(-1).copy()
        pa = np.zeros((po.shape[0], 4), np.int32) if polyaux is None else _i32(polyaux).reshape(-1, 4).copy()
        li = _i32(lights).reshape(-1, 8)
        self.L.ltref_shade_object(_p(v), v.shape[0], _p(n), n.shape[0], _p(po), _p(pa), _p(pv), _p(pn), po.shape[0],
                                  int(pos[0]), int(pos[1]), int(pos[2]), int(ang[0]), int(ang[1]), int(ang[2]),
                                  _p(li), li.shape[0])
        return pv, pa

    def init_object(self, verts, polys, types, polyverts):
        """the real M_init_polygon_object (normals only; at most 1024 vertices, polygons and
        normals) -> (normals int32[nn][3], polynormals int32[records], -2 where a polygon got
        no m_normals array)"""
        v, po, ty = _i32(verts).reshape(-1, 4).copy(), _i32(polys).reshape(-1, 4), _i32(types).reshape(-1)
        pv = _i32(polyverts).reshape(-1, 6).copy()
        assert v.shape[0] <= 1024 and po.shape[0] <= 1024
        normals = np.zeros((1024, 4), np.int32)
        pn = np.full(pv.shape[0], -2, np.int32)
        self.L.ltref_init_object.restype = C.c_int
        nn = self.L.ltref_init_object(_p(v), v.shape[0], _p(po), _p(ty), _p(pv), po.shape[0], _p(normals), _p(pn))
        return normals[:nn, :3].copy(), pn

    def g_line(self, v1, v2, colour, r=0, g=0, b=0):
        """the real G_line on projected {X, Y, Z} end points"""
        a, bb = _i32(v1).copy(), _i32(v2).copy()
        self.L.ltref_g_line(_p(a), _p(bb), int(colour), int(r), int(g), int(b))

    def set_clip_window(self, minx, miny, maxx, maxy):
        """C_init_clipping; stays in force for this library (a process-wide global of the reference)"""
        self.L.ltref_init_clipping(int(minx), int(miny), int(maxx), int(maxy))

    def set_gate(self, indices, verts, log_focus, start):
        """the real M_set_gate from the window `start` -> (return value, window left behind)"""
        idx, v, st = _i32(indices).copy(), _i32(verts).reshape(-1, 4).copy(), _i32(start)
        out = np.zeros(4, np.int32)
        self.L.ltref_set_gate.restype = C.c_int
        rc = self.L.ltref_set_gate(idx.shape[0] - 1, _p(idx), _p(v), int(log_focus), _p(st), _p(out))
        return rc, out

    def bicubic(self, controls, log_render_size, colour, log_texture_space_size, tex, pos, ang, lights, render_type,
                log_focus, force_linear=1):
        """the real M_init_bicubic + M_shade_bicubic + M_render_bicubic of one patch into the current
        frame -> (normals int32[size*size][4] (three words defined), intensities int32[size*size][3])"""
        c, li = _i32(controls).reshape(16, 4), _i32(lights).reshape(-1, 8)
        size = (1 << log_render_size) + 1
        normals = np.zeros((size * size, 4), np.int32)
        inten = np.zeros((size * size, 3), np.int32)
        self.L.ltref_bicubic(_p(c), int(log_render_size), C.c_int(colour), int(log_texture_space_size), int(tex),
                             int(pos[0]), int(pos[1]), int(pos[2]), int(ang[0]), int(ang[1]), int(ang[2]), _p(li), li.shape[0],
                             int(render_type), int(log_focus), int(force_linear), _p(normals), _p(inten))
        return normals, inten

    def g_dot(self, v, colour, r=0, g=0, b=0):
        a = _i32(v).copy()
        self.L.ltref_g_dot(_p(a), C.c_int(colour), int(r), int(g), int(b))

    def g_pixel(self, v, colour):
        """the real G_pixel: the point must be on the screen"""
        a = _i32(v).copy()
        assert 0 <= a[0] < self.w and 0 <= a[1] < self.h
        self.L.ltref_g_pixel(_p(a), C.c_int(colour))

