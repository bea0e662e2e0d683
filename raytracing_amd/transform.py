"""Host-side matrix set-up of the polygon path (32-bit wrap-around integer arithmetic).

Restates, for Python callers of the C-ABI, what the reference does once per object and
frame before the per-vertex loop: T_init_math (Trans.cpp:62-77), T_set_self_rotation
(TransBs.cpp:144-183), T_set_world_rotation (:239-278), M_set_camera
(EngBase.cpp:201-213) and T_cancatinate_self_world (TransBs.cpp:325-361).  The result is
the int32[12] matrix lt_submit_object / lt_transform take.
"""
import math

import numpy as np

T_P = 15


def _w(v):
    """wrap to int32"""
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


def _fx(a, b):
    return _w(a * b) >> T_P


_SIN = [int(math.sin(i / 40.7436642) * (1 << T_P)) for i in range(256)]
_COS = [int(math.cos(i / 40.7436642) * (1 << T_P)) for i in range(256)]


def self_rotation(alp, bet, gam):
    ca, sa, cb, sb, cg, sg = _COS[alp & 255], _SIN[alp & 255], _COS[bet & 255], _SIN[bet & 255], _COS[gam & 255], _SIN[gam & 255]
    sbsg, sbcg = _fx(sb, sg), _fx(sb, cg)
    return [_w(_fx(ca, cg) - _fx(sa, sbsg)), _w(_fx(sa, cg) + _fx(ca, sbsg)), _fx(cb, sg),
            -_fx(sa, cb), _fx(ca, cb), -sb,
            _w(-_fx(ca, sg) - _fx(sa, sbcg)), _w(_fx(ca, sbcg) - _fx(sa, sg)), _fx(cb, cg)]


def world_rotation(alp, bet, gam):
    ca, sa, cb, sb, cg, sg = _COS[alp & 255], _SIN[alp & 255], _COS[bet & 255], _SIN[bet & 255], _COS[gam & 255], _SIN[gam & 255]
    sbsa, sbca = _fx(sb, sa), _fx(sb, ca)
    return [_w(_fx(sg, sbsa) + _fx(cg, ca)), _fx(cb, sa), _w(_fx(sg, ca) - _fx(cg, sbsa)),
            _w(_fx(sg, sbca) - _fx(cg, sa)), _fx(cb, ca), _w(-_fx(cg, sbca) - _fx(sg, sa)),
            -_fx(sg, cb), sb, _fx(cg, cb)]


def make_matrix(cam=(0, 0, 0, 0, 0, 0), obj=(0, 0, 0, 0, 0, 0)):
    """cam = (alp, bet, gam, x, y, z) as passed to M_set_camera; obj = (alp, bet, gam, x, y, z)
    as passed to M_render_polygon_object.  Returns int32[12] rows
    {cx1 cy1 cz1 tx, cx2 cy2 cz2 ty, cx3 cy3 cz3 tz}."""
    w = world_rotation(256 - cam[0], 256 - cam[1], 256 - cam[2])
    s = self_rotation(obj[0], obj[1], obj[2])
    ax, ay, az = _w(obj[3] - cam[3]), _w(obj[4] - cam[4]), _w(obj[5] - cam[5])
    m = []
    for r in range(3):
        wx, wy, wz = w[3 * r], w[3 * r + 1], w[3 * r + 2]
        m += [_w(_fx(s[0], wx) + _fx(s[3], wy) + _fx(s[6], wz)),
              _w(_fx(s[1], wx) + _fx(s[4], wy) + _fx(s[7], wz)),
              _w(_fx(s[2], wx) + _fx(s[5], wy) + _fx(s[8], wz)),
              _w(_fx(ax, wx) + _fx(ay, wy) + _fx(az, wz))]
    return np.array(m, dtype=np.int32)


def self_matrix(alp, bet, gam):
    """T_set_self_rotation -> int32[9] = sx1 sy1 sz1 sx2 sy2 sz2 sx3 sy3 sz3, the matrix
    lt_shade_object takes (M_shade_polygon_object rotates vertices and normals by it)."""
    return np.array(self_rotation(alp, bet, gam), dtype=np.int32)
