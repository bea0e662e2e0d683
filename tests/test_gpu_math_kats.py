"""Known-answer tests of the HIP path's exact-arithmetic helpers (raytracing_amd/csrc/lt_math.h)
on the device, against Python integers: the divisions that go through the floating-point
units must equal the reference's integer arithmetic -- truncating `/` on int32 and, in
T_perspective, on 64-bit operands (TransBs.cpp:425-429); TI_sqrt (TransLn.cpp:38-51);
the 64-bit quotients of T_unit_vector (TransLn.cpp:186-188) -- over each helper's whole
stated operand range.  tests/csrc/lt_kat.hip (built into tests/liblt_kat.so) evaluates
the helpers on arrays."""
import ctypes as C
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
OPS = dict(tdivr=0, udiv_r=1, persp_div=2, tdiv_any=3, ti_sqrt=4, unit_component=5, tdiv=6)
INT_MIN, INT_MAX = -(2**31), 2**31 - 1


@pytest.fixture(scope="module")
def kat():
    import torch
    from raytracing_amd import _lib
    _lib.lib()                                   # the HIP runtime torch ships, first
    so = os.path.join(HERE, "liblt_kat.so")
    assert os.path.exists(so), "tests/liblt_kat.so not built (run __graft_entry__.build())"
    L = C.CDLL(so)
    L.lt_kat_eval.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.lt_kat_eval.restype = C.c_int

    def run(op, a, b=None, c=None):
        a = np.ascontiguousarray(a, dtype=np.int64)
        n = a.size
        d = [None if x is None else torch.from_numpy(np.broadcast_to(np.asarray(x, dtype=np.int64), a.shape).copy()).cuda()
             for x in (a, b, c)]
        out = torch.empty(n, dtype=torch.int64, device="cuda")
        rc = L.lt_kat_eval(OPS[op], *[None if t is None else C.c_void_p(t.data_ptr()) for t in d], C.c_void_p(out.data_ptr()), n)
        assert rc == 0, rc
        return out.cpu().numpy()
    return run


def trunc_div(n, d):
    """C's `/` on exact integers (numpy int64 or object arrays of Python ints)"""
    n = np.asarray(n, dtype=object)
    d = np.asarray(d, dtype=object)
    q = np.abs(n) // np.abs(d)
    return np.where((n < 0) != (d < 0), -q, q)


def wrap32(x):
    x = np.asarray(x, dtype=object)
    return ((x + 2**31) % 2**32) - 2**31


def rand(rng, lo, hi, n):
    return rng.integers(lo, hi, size=n, endpoint=True, dtype=np.int64)


EDGE_N = [INT_MIN, INT_MIN + 1, -1, 0, 1, INT_MAX]
EDGE_D = [1, 2, 3, 255, 16383, 2**15, 2**31 - 1]


def test_tdivr_edges_and_random(kat):
    n, d = np.meshgrid(np.array(EDGE_N + [-255, 254, 255, 256, 65025, -65025, 2**30, -(2**30)], np.int64), np.array(EDGE_D + [7, 254, 256, 65535, 65536, 2**30], np.int64))
    n, d = n.ravel(), d.ravel()
    assert (kat("tdivr", n, d) == trunc_div(n, d).astype(np.int64)).all()
    rng = np.random.default_rng(1)
    for dhi in (16, 255, 2**15, 2**20, INT_MAX):
        n, d = rand(rng, INT_MIN, INT_MAX, 2_000_000), rand(rng, 1, dhi, 2_000_000)
        assert (kat("tdivr", n, d) == trunc_div(n, d).astype(np.int64)).all(), dhi
    # numerators next to multiples of the divisor: where a wrong rounding would show
    d = rand(rng, 1, INT_MAX, 2_000_000)
    q = rand(rng, -(2**31), 2**31 - 1, 2_000_000) // d
    for off in (-1, 0, 1):
        n = np.clip(q * d + off, INT_MIN, INT_MAX)
        assert (kat("tdivr", n, d) == trunc_div(n, d).astype(np.int64)).all(), off


def test_udiv_r_whole_domain_sampled(kat):
    """floor(n / d) for 0 <= n < 2^30, 1 <= d <= 2^15, quotient below 2^15"""
    rng = np.random.default_rng(2)
    d = rand(rng, 1, 2**15, 4_000_000)
    q = rand(rng, 0, 2**15 - 1, 4_000_000)
    for r in (np.zeros_like(d), d - 1, rand(rng, 0, 2**15, 4_000_000) % d):
        n = q * d + r
        keep = n < 2**30
        assert (kat("udiv_r", n[keep], d[keep]) == n[keep] // d[keep]).all()
    # every divisor once with the largest admissible numerator, and small exhaustive corner
    d = np.arange(1, 2**15 + 1, dtype=np.int64)
    n = np.minimum(d * (2**15 - 1) + d - 1, 2**30 - 1)
    assert (kat("udiv_r", n, d) == n // d).all()
    nn, dd = np.meshgrid(np.arange(0, 2048, dtype=np.int64), np.arange(1, 513, dtype=np.int64))
    assert (kat("udiv_r", nn.ravel(), dd.ravel()) == nn.ravel() // dd.ravel()).all()


def test_persp_div_against_reference_vectors_and_python(kat):
    """T_perspective's (int)(((long)x << log_focus) / z): the committed perspective vectors the
    reference build produced (64-bit quotients included), then edge and random operands"""
    k = np.load(os.path.join(HERE, "golden", "stage_kats.npz"), allow_pickle=False)
    t = k["persp_in"].astype(np.int64)
    for lf, key in ((8, "persp_out_lf8"), (11, "persp_out_lf11")):
        want = k[key].astype(np.int64)
        xc, yc = 160 // 2, 120 // 2
        gx = kat("persp_div", t[:, 0], t[:, 2], np.full(t.shape[0], lf))
        gy = kat("persp_div", t[:, 1], t[:, 2], np.full(t.shape[0], lf))
        assert (wrap32(gx + xc).astype(np.int64) == want[:, 0]).all() and (wrap32(gy + yc).astype(np.int64) == want[:, 1]).all()
        gz = kat("tdivr", np.full(t.shape[0], -0x7ffff), t[:, 2])
        assert (gz == want[:, 2]).all()
    rng = np.random.default_rng(3)
    for lf in (0, 1, 7, 10, 13, 16, 24, 30):
        for zhi in (10, 4095, 2**16, INT_MAX):
            x, z = rand(rng, INT_MIN, INT_MAX, 300_000), rand(rng, 10 if zhi > 10 else 1, zhi, 300_000)
            x[:6] = EDGE_N
            want = wrap32(trunc_div(x.astype(object) * (1 << lf), z)).astype(np.int64)
            assert (kat("persp_div", x, z, np.full(x.size, lf)) == want).all(), (lf, zhi)


def test_tdiv_any_and_tdiv(kat):
    n, d = np.meshgrid(np.array(EDGE_N, np.int64), np.array(EDGE_D + [-x for x in EDGE_D] + [INT_MIN], np.int64))
    n, d = n.ravel(), d.ravel()
    want = wrap32(trunc_div(n, d)).astype(np.int64)
    assert (kat("tdiv_any", n, d) == want).all()
    assert (kat("tdiv", n, d) == want).all()
    rng = np.random.default_rng(4)
    for dhi in (300, 2**16, INT_MAX):
        n, d = rand(rng, INT_MIN, INT_MAX, 2_000_000), rand(rng, -dhi, dhi, 2_000_000)
        d[d == 0] = 1
        want = wrap32(trunc_div(n, d)).astype(np.int64)
        assert (kat("tdiv_any", n, d) == want).all(), dhi
        assert (kat("tdiv", n, d) == want).all(), dhi


def test_ti_sqrt(kat):
    """sixteen result bits: floor(sqrt(arg)) while that fits, 0xffff beyond"""
    r = np.arange(0, 0x10000, dtype=np.int64)
    for off in (-1, 0, 1):
        a = np.clip(r * r + off, 0, None)
        want = np.minimum(np.array([math.isqrt(int(v)) for v in a], np.int64), 0xffff)
        assert (kat("ti_sqrt", a) == want).all(), off
    rng = np.random.default_rng(5)
    a = np.concatenate([rand(rng, 0, 2**32, 1_000_000), rand(rng, 2**32, 2**62, 100_000), np.array([0xffff * 0xffff - 1, 0xffff * 0xffff, 2**32 - 1, 2**32, 2**63 - 1], np.int64)])
    want = np.minimum(np.floor(np.sqrt(a.astype(np.float64))).astype(np.int64), 0xffff)
    want = np.where(want * want > a, want - 1, want)            # float sqrt may round up at the top of the range
    assert (kat("ti_sqrt", a) == want).all()


def test_unit_component(kat):
    """(v << 7) / lng, truncating: v any int32, 0 < lng <= 2^20"""
    rng = np.random.default_rng(6)
    v = np.concatenate([np.array(EDGE_N, np.int64), rand(rng, INT_MIN, INT_MAX, 2_000_000), rand(rng, -70000, 70000, 1_000_000)])
    lng = rand(rng, 1, 2**20, v.size)
    lng[:6] = (1, 1, 2**20, 3, 2**20, 1)
    want = wrap32(trunc_div(v.astype(object) * 128, lng)).astype(np.int64)
    assert (kat("unit_component", v, lng) == want).all()
    # multiples of lng and their neighbours
    q = rand(rng, -(2**18), 2**18, 1_000_000)
    for off in (-1, 0, 1):
        vv = np.clip((q * lng[:q.size] + off) // 128 * 1, INT_MIN, INT_MAX)
        want = wrap32(trunc_div(vv.astype(object) * 128, lng[:q.size])).astype(np.int64)
        assert (kat("unit_component", vv, lng[:q.size]) == want).all(), off
