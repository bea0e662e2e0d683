// HIP kernels of the polygon path for gfx950 (MI355X): batched fixed-point vertex
// transform, per-polygon setup (tuples, view-volume test, near-plane clip, perspective,
// back-face cull, screen clip, edge records), tile binning, and the tile fill with an
// LDS-resident depth/colour tile.  Everything is 32-bit integer arithmetic that wraps
// modulo 2^32, like the reference build (int32 "int", arithmetic >>, truncating /).
//
// The reference draws polygon after polygon into one frame with a strict depth compare
// (GraphicsPoly.cpp:357), so the surviving fragment of a pixel is the one with the
// smallest (depth, draw index) pair.  The fill kernel computes exactly that minimum with
// 64-bit LDS atomics on packed (depth, index) keys, which makes the result independent
// of the order in which lanes, waves and workgroups happen to run.  The shading pre-pass
// (per-vertex light, M_shade_polygon_object) is here too.
#include "lt_kernels.h"
#include "lt_math.h"
#include <limits.h>
#include <algorithm>
#include <atomic>

namespace lt {

// ------------------------------------------------------------------ small helpers

// CL_light (Colour.cpp:93-102): per channel (c*min(l,255))/255, evaluated
// arithmetically instead of through the 256x256 table of Colour.cpp:44-56.  A
// negative light is outside the table in the reference (Colour.h:135 clamps only from
// above); here it takes the arithmetic value of the same formula.
__device__ __forceinline__ int cl_light(int colour, int r, int g, int b)
{
    int cr = (colour >> 16) & 0xff, cg = (colour >> 8) & 0xff, cb = colour & 0xff;
    r = r > 255 ? 255 : r;
    g = g > 255 ? 255 : g;
    b = b > 255 ? 255 : b;
    return (((cr * r) / 255) << 16) | (((cg * g) / 255) << 8) | ((cb * b) / 255);
}

template <int DIM>
struct Tup {
    int v[DIM];
};

// element-wise select: keeps tuples in registers (a ?: on the structs themselves is
// compiled as a select between two stack addresses)
template <int DIM>
__device__ __forceinline__ Tup<DIM> sel(bool c, const Tup<DIM> &a, const Tup<DIM> &b)
{
    Tup<DIM> r;
#pragma unroll
    for (int i = 0; i < DIM; i++) r.v[i] = c ? a.v[i] : b.v[i];
    return r;
}

// ------------------------------------------------------------------ transform

struct Mat12 {
    int m[12];
};

// T_cancatinated_rotation, TransBs.cpp:373-393: one vertex per lane, 16-byte load,
// 12-byte store (the 4th lane of the destination stays as it was).
__global__ void __launch_bounds__(256) transform_kernel(Mat12 M, const int4 *__restrict__ from,
                                                        int *__restrict__ to, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int4 p = from[i];
    int o[3];
#pragma unroll
    for (int r = 0; r < 3; r++)
        o[r] = wadd(wadd(wadd(wmul(M.m[4 * r], p.x) >> 15, wmul(M.m[4 * r + 1], p.y) >> 15),
                         wmul(M.m[4 * r + 2], p.z) >> 15),
                    M.m[4 * r + 3]);
    int *d = to + (size_t)i * 4;
    d[0] = o[0];
    d[1] = o[1];
    d[2] = o[2];
}

// ------------------------------------------------------------------ shading pre-pass

// T_norm, TransLn.cpp:75-103
__device__ __forceinline__ long long t_norm(const int (&v)[3])
{
    if (v[0] > 0xffff || v[1] > 0xffff || v[2] > 0xffff || v[0] < -0xffff || v[1] < -0xffff || v[2] < -0xffff)
        return (long long)(ti_sqrt((unsigned long long)((((long long)v[0]) >> 8) * v[0] + (((long long)v[1]) >> 8) * v[1] +
                                                        (((long long)v[2]) >> 8) * v[2])) << 4);
    return (long long)ti_sqrt((unsigned long long)((long long)v[0] * v[0] + (long long)v[1] * v[1] + (long long)v[2] * v[2]));
}

// T_self_rotation, TransBs.cpp:194-214
__device__ __forceinline__ void self_rotate(const int (&s)[9], const int4 p, int (&o)[3])
{
#pragma unroll
    for (int r = 0; r < 3; r++)
        o[r] = wadd(wadd(wmul(s[3 * r], p.x) >> 15, wmul(s[3 * r + 1], p.y) >> 15), wmul(s[3 * r + 2], p.z) >> 15);
}

// T_scalar_product, TransLn.cpp:201-208
__device__ __forceinline__ int scalar_product(const int (&a)[3], const int (&b)[3])
{
    return (int)(((long long)a[0] * b[0] + (long long)a[1] * b[1] + (long long)a[2] * b[2]) >> 7);
}

// T_normal_plane (TransLn.cpp:150-171) for every polygon of an object, as the first loop of
// M_init_polygon_object runs it (EngPoly.cpp:392-398): the unit normal, 7 fraction bits, of
// the plane through the vertices of the polygon's first three records.  The products of the
// vector product wrap in 32 bits like the reference's int arithmetic.  One polygon per
// lane; out[poly] = {nx, ny, nz, 0}.  A zero-length product (where the reference stops with
// "Found a zero length vector") or records/indices out of range give {0, 0, 0, 0} and are
// counted.
__global__ void __launch_bounds__(256) plane_normals_kernel(const int *__restrict__ verts, int nv,
                                                            const int *__restrict__ polys, int np,
                                                            const int *__restrict__ polyverts, int npv,
                                                            int4 *__restrict__ out, Status *status)
{
    const int poly = blockIdx.x * blockDim.x + threadIdx.x;
    bool zero = false;
    if (poly < np) {
        const int4 pd = ((const int4 *)polys)[poly];
        const int first = pd.w;
        int4 r = make_int4(0, 0, 0, 0);
        zero = true;
        if (first >= 0 && (long long)first + 3 <= (long long)npv) {
            const int i0 = polyverts[(size_t)first * 6], i1 = polyverts[(size_t)(first + 1) * 6], i2 = polyverts[(size_t)(first + 2) * 6];
            if ((unsigned)i0 < (unsigned)nv && (unsigned)i1 < (unsigned)nv && (unsigned)i2 < (unsigned)nv) {
                const int4 a = ((const int4 *)verts)[i0], b = ((const int4 *)verts)[i1], c = ((const int4 *)verts)[i2];
                const int v[3] = {wsub(b.x, a.x), wsub(b.y, a.y), wsub(b.z, a.z)};
                const int w[3] = {wsub(c.x, a.x), wsub(c.y, a.y), wsub(c.z, a.z)};
                const int n[3] = {wsub(wmul(v[1], w[2]), wmul(v[2], w[1])), wsub(wmul(v[2], w[0]), wmul(v[0], w[2])),
                                  wsub(wmul(v[0], w[1]), wmul(v[1], w[0]))};
                const long long lng = t_norm(n);
                if (lng != 0) {
                    r = make_int4(unit_component(n[0], lng), unit_component(n[1], lng), unit_component(n[2], lng), 0);
                    zero = false;
                }
            }
        }
        out[poly] = r;
    }
    const unsigned long long m = __ballot(zero);
    if (m && (threadIdx.x & 63) == (__ffsll((long long)m) - 1))
        atomicAdd(&status[blockIdx.x % STATUS_SLOTS].zero_light_vectors, (unsigned long long)__popcll(m));
}

// M_shade_vertex, EngBase.cpp:220-288 (_RGB_): light from the list at a world-space point
// with a world-space normal.  `zero` counts point lights at zero distance (left out; the
// reference stops in T_unit_vector there).
__device__ __forceinline__ void shade_vertex(const ShadeArgs &a, const int (&wv)[3], const int (&wn)[3], int (&rgb)[3],
                                             unsigned &zero)
{
    rgb[0] = 0; rgb[1] = 0; rgb[2] = 0;
    for (int j = 0; j < a.nlights; j++) {
        const int type = a.lights[j][0];
        const int lc[3] = {a.lights[j][5], a.lights[j][6], a.lights[j][7]};
        const int lp[3] = {a.lights[j][1], a.lights[j][2], a.lights[j][3]};
        int prd = -1;
        if (type == 1) {                                   // M_AMBIENT
#pragma unroll
            for (int q = 0; q < 3; q++) rgb[q] = wadd(rgb[q], lc[q]);
        } else if (type == 2) {                            // M_POINT: T_unit_vector, TransLn.cpp:175-190
            const int v[3] = {wsub(lp[0], wv[0]), wsub(lp[1], wv[1]), wsub(lp[2], wv[2])};
            const long long lng = t_norm(v);
            if (lng == 0) zero++;
            else {
                const int lv[3] = {unit_component(v[0], lng), unit_component(v[1], lng), unit_component(v[2], lng)};
                prd = scalar_product(lv, wn);
            }
        } else if (type == 4) {                            // M_DIRECT
            prd = (int)(0u - (unsigned)scalar_product(lp, wn));
        }
        if (type != 1 && prd >= 0) {
#pragma unroll
            for (int q = 0; q < 3; q++) rgb[q] = wadd(rgb[q], wmul(prd, lc[q]) >> 7);
        }
    }
}

// The shading loop of M_shade_bicubic (EngBcub.cpp:314-325): n points already in world space,
// their normals rotated by the self rotation here (T_self_rotation, EngBcub.cpp:291);
// rgb[i] = {R, G, B, 0}.  One point per lane.
__global__ void __launch_bounds__(256) shade_points_kernel(ShadeArgs a, const int4 *__restrict__ points,
                                                           const int4 *__restrict__ normals, int n, int4 *__restrict__ rgb_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned zero = 0;
    if (i < n) {
        const int4 p = points[i];
        const int wv[3] = {p.x, p.y, p.z};
        int wn[3], rgb[3];
        self_rotate(a.s, normals[i], wn);
        shade_vertex(a, wv, wn, rgb, zero);
        rgb_out[i] = make_int4(rgb[0], rgb[1], rgb[2], 0);
    }
    unsigned long long zs = zero;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) zs += __shfl_down(zs, d);
    if ((threadIdx.x & 63) == 0 && zs) atomicAdd(&a.status[blockIdx.x % STATUS_SLOTS].zero_light_vectors, zs);
}

// M_shade_polygon over M_shade_vertex (EngBase.cpp:220-364): one polygon per lane; every
// record of the polygon gets its R,G,B from the light list, the polygon its flat
// illumination (the truncated mean over its n_edges+1 records).  Vertices and normals are
// moved to world space as they are read (T_self_rotation + T_translation of
// M_shade_polygon_object, EngPoly.cpp:476-479).
__global__ void __launch_bounds__(256) shade_kernel(ShadeArgs a)
{
    const int poly = blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    unsigned zero = 0;
    if (poly < a.np) {
        const int4 pd = ((const int4 *)a.polys)[poly];
        const int e = pd.z;
        const long long first = pd.w;
        if (e < 0 || first < 0 || first + e + 1 > (long long)a.npv) bad = true;
        for (int k = 0; !bad && k <= e; k++) {
            const int vi = a.polyverts[(first + k) * 6], ni = a.polynormals[first + k];
            if ((unsigned)vi >= (unsigned)a.nv || (unsigned)ni >= (unsigned)a.nn) bad = true;
        }
        if (!bad) {
            int sum[3] = {0, 0, 0};
            // four records at a time, their loads in flight together (a triangle is one round)
            for (int k0 = 0; k0 <= e; k0 += 4) {
                // whole records are read and written back whole: storing only R,G,B would
                // leave holes in every line, which costs more than the extra bytes
                int2 ra[4], rc[4];
                int ni[4];
                int4 pv[4], pn[4];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const bool on = k0 + g <= e;
                    const int2 *rec = (const int2 *)(a.polyverts + (first + k0 + (on ? g : 0)) * 6);
                    ra[g] = rec[0];
                    rc[g] = rec[2];
                    ni[g] = a.polynormals[first + k0 + (on ? g : 0)];
                }
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    pv[g] = ((const int4 *)a.verts)[ra[g].x];
                    pn[g] = ((const int4 *)a.normals)[ni[g]];
                }
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    if (k0 + g > e) continue;
                    int2 *rec = (int2 *)(a.polyverts + (first + k0 + g) * 6);
                    int wv[3], wn[3];
                    self_rotate(a.s, pv[g], wv);
                    wv[0] = wadd(wv[0], a.x); wv[1] = wadd(wv[1], a.y); wv[2] = wadd(wv[2], a.z);
                    self_rotate(a.s, pn[g], wn);
                    int rgb[3];
                    shade_vertex(a, wv, wn, rgb, zero);
                    rec[0] = make_int2(ra[g].x, rgb[0]);
                    rec[1] = make_int2(rgb[1], rgb[2]);
                    rec[2] = rc[g];
#pragma unroll
                    for (int q = 0; q < 3; q++) sum[q] = wadd(sum[q], rgb[q]);
                }
            }
            int4 *ax = (int4 *)a.polyaux + poly;
            const double rn = drcp(e + 1);
            *ax = make_int4(tdivr(sum[0], rn), tdivr(sum[1], rn), tdivr(sum[2], rn), ax->w);      // whole 16 bytes, 4th kept
        }
    }
    // counters: one atomic per wave
    Status *st = a.status + (blockIdx.x % STATUS_SLOTS);
    const unsigned long long bm = __ballot(bad);
    if (bm && (threadIdx.x & 63) == (__ffsll((long long)bm) - 1)) atomicAdd(&st->overflow_polys, (unsigned long long)__popcll(bm));
    unsigned long long zs = zero;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) zs += __shfl_down(zs, d);
    if ((threadIdx.x & 63) == 0 && zs) atomicAdd(&st->zero_light_vectors, zs);
}

// Lanes of one wave exchange data through LDS: the wave's LDS operations complete in issue
// order, so all this needs is that the compiler keeps the order (a release fence limited
// to LDS: a full fence would also wait for every global access in flight, prefetches
// included).
#define LDS_WAVE_SYNC()                                                    \
    do {                                                                   \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");    \
        __builtin_amdgcn_wave_barrier();                                   \
    } while (0)

// lane `l` of v, l wave-uniform
__device__ __forceinline__ int rdlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

// The same transform applied to one vertex as the setup kernel reads it, so that an
// object's view-space vertices never make a round trip through HBM.
__device__ __forceinline__ int4 view_vertex(const SetupArgs &a, int4 p)
{
    if (a.use_matrix) {
        int o[3];
#pragma unroll
        for (int r = 0; r < 3; r++)
            o[r] = wadd(wadd(wadd(wmul(a.m[4 * r], p.x) >> 15, wmul(a.m[4 * r + 1], p.y) >> 15),
                             wmul(a.m[4 * r + 2], p.z) >> 15),
                        a.m[4 * r + 3]);
        p.x = o[0]; p.y = o[1]; p.z = o[2];
    }
    return p;
}

__device__ __forceinline__ int4 load_vertex(const SetupArgs &a, int idx)
{
    return view_vertex(a, ((const int4 *)a.verts)[idx]);
}

// Where a polygon's vertices come from.  One object per batch: the whole vertex array and the
// batch's matrix (kernel arguments).  A table of objects (MULTI): the polygon's object gives
// the range of the shared vertex array that its records index and the matrix, which is read
// from the table with every vertex (the general set-up kernels are short of registers; the
// table is a few cache lines).
template <bool MULTI>
struct VertexSource {
    int obj, vbase;
    unsigned vcount;
    // false: the polygon names no object of the table
    __device__ __forceinline__ bool init(const SetupArgs &a, int poly)
    {
        obj = 0; vbase = 0; vcount = (unsigned)a.nv;
        if (MULTI) {
            obj = a.polyobj[poly];
            if ((unsigned)obj >= (unsigned)a.nobj) { obj = 0; vcount = 0; return false; }
            const int4 r = ((const int4 *)(a.objects + obj))[3];
            vbase = r.x; vcount = (unsigned)r.y;
            if (vbase < 0 || r.y < 0 || (long long)vbase + r.y > (long long)a.nv) { vcount = 0; return false; }
        }
        return true;
    }
    __device__ __forceinline__ bool valid(int idx) const { return (unsigned)idx < vcount; }
    __device__ __forceinline__ int4 load(const SetupArgs &a, int idx) const
    {
        if (!MULTI) return load_vertex(a, idx);
        int4 p = ((const int4 *)a.verts)[vbase + idx];
        const int4 *m = (const int4 *)(a.objects + obj);
        int o[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const int4 row = m[r];
            o[r] = wadd(wadd(wadd(wmul(row.x, p.x) >> 15, wmul(row.y, p.y) >> 15), wmul(row.z, p.z) >> 15), row.w);
        }
        p.x = o[0]; p.y = o[1]; p.z = o[2];
        return p;
    }
};

// ------------------------------------------------------------------ clipping

template <int DIM>
__device__ __forceinline__ void midpoint(Tup<DIM> &m, const Tup<DIM> &l, const Tup<DIM> &r)
{
#pragma unroll
    for (int i = 0; i < DIM; i++) m.v[i] = wadd(l.v[i], r.v[i]) >> 1;
}

// C_line_x_clipping / C_line_y_clipping (Clipp2D.cpp:94-228; BOTH sides) and
// C_line_z_clipping (Clipp3D.cpp:70-110; near side only): binary search on every
// dimension of the tuple until coordinate AX equals the bound.  `moved` restates
// C_2D_clipping / C_3D_clipping.  The search halves an integer interval, so it ends
// within 32 steps unless l+r overflows; `bad` reports a search that did not end.
template <int DIM, int AX, bool BOTH>
__device__ __forceinline__ bool clip_line(const Tup<DIM> &v1, const Tup<DIM> &v2, int bmin, int bmax,
                                          Tup<DIM> &o1, Tup<DIM> &o2, bool &moved, bool &bad)
{
    bool swap = !(v1.v[AX] < v2.v[AX]);
    Tup<DIM> lo = sel(swap, v2, v1), hi = sel(swap, v1, v2);
    moved = false;
    if (BOTH) {
        if (lo.v[AX] >= bmax || hi.v[AX] <= bmin) return false;
    } else {
        if (hi.v[AX] <= bmin) return false;
    }
    // The reference keeps the interval's two ends l, r and its midpoint m; a step replaces the end on m's
    // side by m (l if m is below the bound, `whereto`, else r) and halves again.  The end that stays is
    // other = whereto ? r : l, the new midpoint is (m + other) >> 1 whichever end that was, and the end
    // that stays in the step after is `other` again if the search goes the same way twice, else the old
    // midpoint: one select a dimension and step instead of two, the same values in the same order.
    if (lo.v[AX] <= bmin) {
        Tup<DIM> m = lo, other = hi;
        bool whereto = true;
        int guard = 0;
        while (m.v[AX] != bmin) {
            Tup<DIM> next;
            midpoint(next, m, other);
            const bool w2 = next.v[AX] < bmin;
            other = sel(w2 == whereto, other, m);
            m = next;
            whereto = w2;
            if (++guard > 40) { bad = true; break; }
        }
        lo = m;
        moved = !swap;
    }
    if (BOTH && hi.v[AX] >= bmax) {
        Tup<DIM> m = hi, other = lo;
        bool whereto = false;
        int guard = 0;
        while (m.v[AX] != bmax) {
            Tup<DIM> next;
            midpoint(next, m, other);
            const bool w2 = next.v[AX] < bmax;
            other = sel(w2 == whereto, other, m);
            m = next;
            whereto = w2;
            if (++guard > 40) { bad = true; break; }
        }
        hi = m;
        moved = moved || swap;
    }
    o1 = sel(swap, hi, lo);
    o2 = sel(swap, lo, hi);
    return true;
}

// MI_tmapping_vectors (EngBase.cpp:145-179): texture orientation vectors u, v from the
// polygon's first three tuples (view space, before clipping).  type = m_type & M_QUAD.
// The general case is T_linear_solve (TransBs.cpp:440-515, fixed-point branch) for the
// 2x2 system with three right-hand sides, log_source_multiplyer = 16.  Returns false
// where the reference would divide by zero.
__device__ bool tmapping_vectors(int type, const Tup<8> &p1, const Tup<8> &p2, const Tup<8> &p3, int log_space,
                                 int (&u)[3], int (&v)[3])
{
    if (type) {
        if (type & 4) {        // M_QUAD_XY
#pragma unroll
            for (int j = 0; j < 3; j++) { u[j] = wsub(p2.v[j], p1.v[j]); v[j] = wsub(p3.v[j], p2.v[j]); }
        } else {
#pragma unroll
            for (int j = 0; j < 3; j++) { u[j] = wsub(p1.v[j], p2.v[j]); v[j] = wsub(p2.v[j], p3.v[j]); }
        }
        return true;
    }
    int a00 = wsub(p2.v[6], p1.v[6]) >> 16, a01 = wsub(p2.v[7], p1.v[7]) >> 16;
    int a10 = wsub(p3.v[6], p2.v[6]) >> 16, a11 = wsub(p3.v[7], p2.v[7]) >> 16;
    int b0[3], b1[3];
#pragma unroll
    for (int j = 0; j < 3; j++) { b0[j] = wsub(p2.v[j], p1.v[j]); b1[j] = wsub(p3.v[j], p2.v[j]); }
    // pivot search over column 0 (TransBs.cpp:477-481)
    int mx = 0, num = 0;
    { int pv = a00 >= 0 ? a00 : -a00; if (pv > mx) { mx = pv; num = 0; } }
    { int pv = a10 >= 0 ? a10 : -a10; if (pv > mx) { mx = pv; num = 1; } }
    if (mx != 0) {
        if (num != 0) {
            int t;
            t = a00; a00 = a10; a10 = t;
            t = a01; a01 = a11; a11 = t;
#pragma unroll
            for (int j = 0; j < 3; j++) { t = b0[j]; b0[j] = b1[j]; b1[j] = t; }
        }
        a11 = wsub(a11, tdiv_any(wmul(a01, a10), a00));
#pragma unroll
        for (int j = 0; j < 3; j++) b1[j] = wsub(b1[j], tdiv_any(wmul(b0[j], a10), a00));
    }
    if (a11 == 0 || a00 == 0) return false;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        int x1 = tdiv_any(wshl(b1[j], log_space), a11);
        int sum = wmul(a01, x1) >> log_space;
        int x0 = tdiv_any(wshl(wsub(b0[j], sum), log_space), a00);
        u[j] = x0;
        v[j] = x1;
    }
    return true;
}

// ------------------------------------------------------------------ setup

constexpr int KIND_FLAT = 2, KIND_SHADED = 4, KIND_TEX_LIN = 8, KIND_TEX_PRP = 9;

// Counting half of the binning done where the header is made (a.tile_count != null): one
// add per touched tile.
__device__ __forceinline__ void count_tiles(const SetupArgs &a, int by0, int by1, int minx, int maxx)
{
    const int tx0 = minx >> a.tile_w_log, tx1 = maxx >> a.tile_w_log, ty0 = by0 / TH, ty1 = (by1 - 1) / TH;
    for (int ty = ty0; ty <= ty1; ty++)
        for (int tx = tx0; tx <= tx1; tx++) atomicAdd(a.tile_count + ty * a.tiles_x + tx, 1u);
}

// 16-byte piece `piece` of the record of edge `e` of polygon `poly`.  Edges 0..2 of a
// polygon lie together in one record of record_units() pieces: unit 0 is a copy of the
// header, units 1..3 the depth pieces (piece 0) of the three edges, so that the depth pass
// of a triangle reads one 64-byte run; the other pieces follow edge by edge.  Edges 3 and up
// are planes [edge][piece][polygon] behind the records.
__host__ __device__ constexpr int record_units(int npieces) { return 4 + 3 * (npieces - 1); }
__device__ __forceinline__ size_t edge_unit(int np, int npieces, int e, int piece, int poly)
{
    if (e < 3) return (size_t)poly * record_units(npieces) + (piece == 0 ? 1 + e : 4 + e * (npieces - 1) + (piece - 1));
    return (size_t)np * record_units(npieces) + ((size_t)((e - 3) * npieces + piece) * np + poly);
}
__device__ __forceinline__ int4 *edge_piece(int *edges, int np, int npieces, int e, int piece, int poly)
{
    return (int4 *)edges + edge_unit(np, npieces, e, piece, poly);
}
__device__ __forceinline__ const int4 *edge_piece(const int *edges, int np, int npieces, int e, int piece, int poly)
{
    return (const int4 *)edges + edge_unit(np, npieces, e, piece, poly);
}

// The deltas of an edge as every later stage wants them: |dx| | |dy| << 15 | (dx >= 0) << 30 |
// (dy < 0) << 31 (coordinates are below 2^15: the screen is at most 16384 wide and high).
__device__ __forceinline__ int edge_pack(int x1, int y1, int x2, int y2)
{
    const int dx = x2 - x1, dy = y2 - y1;
    const int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
    return adx | (ady << 15) | (dx >= 0 ? 1 << 30 : 0) | (dy < 0 ? (int)0x80000000 : 0);
}

// Edge record of one scanned edge v1 -> v2 (after the y clip): GI_scan's set-up,
// GraphicsPoly.cpp:147-176.  record = {x1|y1<<16, edge_pack(deltas), Zstart, Zinc, other starts..,
// other incs..} in 16-byte pieces, placed by edge_piece().  (x2, y2 themselves are not kept: every reader wants the deltas.)
template <int DIM>
__device__ __forceinline__ void make_edge_record(const Tup<DIM> &v1, const Tup<DIM> &v2, int (&rec)[edge_ints(DIM - 2)])
{
    constexpr int NA = DIM - 2;
    constexpr int ESZ = edge_ints(NA);
    const int x1 = v1.v[0], y1 = v1.v[1], x2 = v2.v[0], y2 = v2.v[1];
    int dx = x2 - x1, dy = y2 - y1;
    dx = dx < 0 ? -dx : dx;
    dy = dy < 0 ? -dy : dy;
    const int L = dx > dy ? dx : dy;
#pragma unroll
    for (int q = 0; q < ESZ; q++) rec[q] = 0;
    rec[0] = x1 | (y1 << 16);
    rec[1] = edge_pack(x1, y1, x2, y2);
    const double rL = drcp(L > 0 ? L : 1);
    rec[2] = wshl(v1.v[2], 16);
    rec[3] = L > 0 ? tdivr(wshl(wsub(v2.v[2], v1.v[2]), 16), rL) : 0;
#pragma unroll
    for (int q = 1; q < NA; q++) {
        rec[3 + q] = v1.v[2 + q];
        rec[3 + (NA - 1) + q] = L > 0 ? tdivr(wsub(v2.v[2 + q], v1.v[2 + q]), rL) : 0;
    }
}

template <int DIM>
__device__ __forceinline__ void store_edge_record(const SetupArgs &a, int poly, int slot, const Tup<DIM> &v1,
                                                  const Tup<DIM> &v2)
{
    constexpr int ESZ = edge_ints(DIM - 2);
    int rec[ESZ];
    make_edge_record<DIM>(v1, v2, rec);
#pragma unroll
    for (int q = 0; q < ESZ / 4; q++)
        *edge_piece(a.edges, a.np, ESZ / 4, slot, q, poly) = make_int4(rec[4 * q], rec[4 * q + 1], rec[4 * q + 2], rec[4 * q + 3]);
}

// The header copy at the front of a polygon's record (what the fill reads).
__device__ __forceinline__ int4 *record_header(int *edges, int npieces, int poly)
{
    return (int4 *)edges + (size_t)poly * record_units(npieces);
}
__device__ __forceinline__ const int4 *record_header(const int *edges, int npieces, int poly)
{
    return (const int4 *)edges + (size_t)poly * record_units(npieces);
}

// Triangles that need no clipping at all take this kernel: volume test passed with every
// z >= 10, and all four projected tuples strictly inside the screen, so that the z, x and y
// clippers would pass every edge through unchanged (Clipp3D.cpp:91, Clipp2D.cpp:115-136,
// 188-209).  The screen clipper then emits the second tuple of each edge, i.e. the
// polygon rotated by one (Clipp2D.cpp:265-272): scanned edges are p1->p2, p2->p3, p3->p1
// with p3 the stored closing tuple.  Everything else (other edge counts, near-plane or
// screen-border contact, perspective-mapped textures) is appended to a queue for the
// general kernel, which then runs only on those polygons with all its lanes busy.
// (Flat and Gouraud: held to the 96 registers of five waves a SIMD -- it took 98 when left alone --, because its waves share a
// SIMD's registers with three waves of a running fill: see raster_kernel.  The textured kernel would spill.)
template <int DIM, bool MULTI>
__global__ void __launch_bounds__(SETUP_FAST_THREADS) __attribute__((amdgpu_waves_per_eu(DIM <= 6 ? 5 : 4, DIM <= 6 ? 5 : 4))) setup_fast_kernel(SetupArgs a)
{
    // The kernels of a batch's set-up chain run beside the fills of earlier batches, each waiting
    // for the one before it: their waves take the highest wave priority, so that a SIMD shared
    // with fill waves (which rotate through the priorities) serves the chain first.
    __builtin_amdgcn_s_setprio(3);
    constexpr int NA = DIM - 2;
    constexpr int ESZ = edge_ints(NA);
    constexpr int RU = record_units(ESZ / 4);
    // per wave: 64 x 4 vertex records of 24 bytes on the way in, 32 polygon records on the way out
    constexpr int STAGE_UNITS = 6 * 64 > 32 * RU ? 6 * 64 : 32 * RU;
    __shared__ int4 stage_f[SETUP_FAST_THREADS / 64][STAGE_UNITS];
    const int poly = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int4 *stg = stage_f[threadIdx.x >> 6];
    const bool inrange = poly < a.np;
    const bool wave_full = __ballot(inrange) == ~0ull;
    bool slow = false, drawn = false, vary = false, offscreen = false, visible = false, straddles = false;
    Tup<DIM> t[4];
    int colour = 0, tex = 0, kind = a.render_type, by0 = 0, by1 = 0, minx = 0, maxx = 0;
    if (inrange) {
        const int4 pd = ((const int4 *)a.polys)[poly];
        colour = pd.x; tex = pd.y;
        const int e = pd.z, first = pd.w;
        if (kind == KIND_TEX_LIN && tex < 0) kind = KIND_SHADED;
        if (e != 3 || e > a.max_edges || first < 0 || (long long)first + 4 > (long long)a.npv ||
            (kind == KIND_TEX_LIN && tex >= 64))
            slow = true;
        // a table of objects: this polygon's matrix and its range of the vertex array
        int om[12];
        int vbase = 0;
        unsigned vcount = (unsigned)a.nv;
        if (MULTI) {
            const int o = a.polyobj[poly];
            int4 r0 = make_int4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
            if ((unsigned)o < (unsigned)a.nobj) {
                const int4 *od = (const int4 *)(a.objects + o);
                r0 = od[0]; r1 = od[1]; r2 = od[2]; r3 = od[3];
            } else {
                slow = true;             // the general kernel reports it
            }
            om[0] = r0.x; om[1] = r0.y; om[2] = r0.z; om[3] = r0.w;
            om[4] = r1.x; om[5] = r1.y; om[6] = r1.z; om[7] = r1.w;
            om[8] = r2.x; om[9] = r2.y; om[10] = r2.z; om[11] = r2.w;
            vbase = r3.x; vcount = (unsigned)r3.y;
            if (vbase < 0 || r3.y < 0 || (long long)vbase + r3.y > (long long)a.nv) slow = true;
        }
        int2 ra[4] = {}, rb[4] = {}, rc[4] = {};
        // A wave whose 64 triangles keep their vertex records one after the other
        // (first = first0 + 4*lane) reads the 6 KB block with consecutive 16-byte loads
        // and hands each lane its 96 bytes through LDS: a lane-per-record gather would
        // touch 64 different cache lines per load instruction.
        const int first0 = rdlane(first, 0);
        const bool rec_seq = wave_full && __ballot(!slow && first == first0 + 4 * lane) == ~0ull && (first0 & 1) == 0 && ((size_t)a.polyverts & 15) == 0 &&
                             (long long)first0 + 256 <= (long long)a.npv;
        if (rec_seq) {
            const int4 *blk = (const int4 *)(a.polyverts + (size_t)first0 * 6);
#pragma unroll
            for (int k = 0; k < 6; k++) stg[k * 64 + lane] = blk[k * 64 + lane];
            LDS_WAVE_SYNC();
            int r[24];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const int4 q = stg[lane * 6 + k];
                r[4 * k] = q.x; r[4 * k + 1] = q.y; r[4 * k + 2] = q.z; r[4 * k + 3] = q.w;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 4; i++) {
                ra[i] = make_int2(r[6 * i], r[6 * i + 1]);
                rb[i] = make_int2(r[6 * i + 2], r[6 * i + 3]);
                rc[i] = make_int2(r[6 * i + 4], r[6 * i + 5]);
            }
        } else if (!slow) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int2 *rec = (const int2 *)(a.polyverts + (size_t)(first + i) * 6);
                ra[i] = rec[0];
                rb[i] = DIM >= 6 ? rec[1] : make_int2(0, 0);
                rc[i] = DIM == 8 ? rec[2] : make_int2(0, 0);
            }
        }
        if (!slow) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if ((unsigned)ra[i].x >= vcount) slow = true;
                ra[i].x += vbase;                 // index into the shared vertex array
            }
        }
        // likewise the vertices of a triangle soup: indices v0 + 3*lane + {0, 1, 2, 0}
        const int v0 = rdlane(ra[0].x, 0);
        const bool vert_seq = wave_full && (long long)v0 + 192 <= (long long)a.nv && v0 >= 0 &&
                              __ballot(!slow && ra[0].x == v0 + 3 * lane && ra[1].x == ra[0].x + 1 &&
                                       ra[2].x == ra[0].x + 2 && ra[3].x == ra[0].x) == ~0ull;
        int4 pv[4];
        if (vert_seq) {
            const int4 *blk = (const int4 *)a.verts + v0;
#pragma unroll
            for (int k = 0; k < 3; k++) stg[k * 64 + lane] = blk[k * 64 + lane];
            LDS_WAVE_SYNC();
#pragma unroll
            for (int i = 0; i < 3; i++) pv[i] = stg[lane * 3 + i];
            pv[3] = pv[0];
            __builtin_amdgcn_wave_barrier();
        } else if (!slow) {
#pragma unroll
            for (int i = 0; i < 4; i++) pv[i] = ((const int4 *)a.verts)[ra[i].x];
        }
        // The fourth tuple of a triangle closes it: as a rule it names the first vertex again.  Where it
        // does so in every lane of the wave, its position -- view space, then projected -- is the first
        // tuple's and is not worked out a second time (a quarter of the kernel's divisions).
        const bool closes = __ballot(!slow && ra[3].x != ra[0].x) == 0ull;
        if (!slow) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                int4 p = pv[i];
                if (i == 3 && closes) {
                    p.x = t[0].v[0]; p.y = t[0].v[1]; p.z = t[0].v[2];
                } else
                if (MULTI) {
                    int o3[3];
#pragma unroll
                    for (int r = 0; r < 3; r++)
                        o3[r] = wadd(wadd(wadd(wmul(om[4 * r], p.x) >> 15, wmul(om[4 * r + 1], p.y) >> 15), wmul(om[4 * r + 2], p.z) >> 15), om[4 * r + 3]);
                    p.x = o3[0]; p.y = o3[1]; p.z = o3[2];
                } else {
                    p = view_vertex(a, p);
                }
                t[i].v[0] = p.x; t[i].v[1] = p.y; t[i].v[2] = p.z;
                if (DIM >= 6) { t[i].v[3] = wshl(ra[i].y, 16); t[i].v[4] = wshl(rb[i].x, 16); t[i].v[5] = wshl(rb[i].y, 16); }
                if (DIM == 8) {
                    const bool tx = kind == KIND_TEX_LIN;
                    t[i].v[DIM - 2] = tx ? wshl(rc[i].x, 16) : 0;
                    t[i].v[DIM - 1] = tx ? wshl(rc[i].y, 16) : 0;
                }
            }
        }
        bool go = !slow;
        if (go) {
            // C_volume_clipping, Clipp3D.cpp:46-58
            int mn[3], mx[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                mn[j] = min(min(t[0].v[j], t[1].v[j]), min(t[2].v[j], t[3].v[j]));
                mx[j] = max(max(t[0].v[j], t[1].v[j]), max(t[2].v[j], t[3].v[j]));
            }
            if ((mx[2] < mn[0]) || (mx[2] < mn[1]) || (mx[2] < -mx[0]) || (mx[2] < -mx[1]) || (mx[2] <= 10)) go = false;
            else if (mn[2] < 10) { slow = true; go = false; }
        }
        if (go && DIM == 8 && kind == KIND_TEX_LIN && (a.force_linear || t[0].v[2] < 1024)) { slow = true; go = false; }
        if (go) {
            if (DIM == 8)
                vary = t[1].v[3] != t[0].v[3] || t[1].v[4] != t[0].v[4] || t[1].v[5] != t[0].v[5] ||
                       t[2].v[3] != t[0].v[3] || t[2].v[4] != t[0].v[4] || t[2].v[5] != t[0].v[5] ||
                       t[3].v[3] != t[0].v[3] || t[3].v[4] != t[0].v[4] || t[3].v[5] != t[0].v[5];
            // T_perspective, TransBs.cpp:425-432
            const int xc = a.W / 2, yc = a.H / 2;
            bool inside = true, left = true, right = true, above = true, below = true;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (i == 3 && closes) {         // the first tuple's projection; the tests below have seen it
                    t[3].v[0] = t[0].v[0]; t[3].v[1] = t[0].v[1]; t[3].v[2] = t[0].v[2];
                    continue;
                }
                const int z = t[i].v[2];
                const double rz = drcp(z);
                t[i].v[0] = wadd(persp_div(t[i].v[0], a.log_focus, z, rz), xc);
                t[i].v[1] = wadd(persp_div(t[i].v[1], a.log_focus, z, rz), yc);
                t[i].v[2] = tdivr(-0x7ffff, rz);
                inside = inside && t[i].v[0] > a.cx0 && t[i].v[0] < a.cx1 && t[i].v[1] > a.cy0 && t[i].v[1] < a.cy1;
                left = left && t[i].v[0] <= a.cx0;
                right = right && t[i].v[0] >= a.cx1;
                above = above && t[i].v[1] <= a.cy0;
                below = below && t[i].v[1] >= a.cy1;
            }
            // Wholly on the outer side of one screen border: the x clipper drops every
            // edge whose two ends are at or beyond the same bound (Clipp2D.cpp:115-118),
            // and so does the y clipper of the scan (:188-191) -- clipped points lie
            // between their edge's ends -- so nothing is drawn; the polygon still counts
            // as drawn if it faces the viewer, as in M_render_polygon.
            offscreen = left || right || above || below;
            // neither: the polygon meets a screen border and goes, projected, to the screen-level
            // set-up (x and y clipping) if it faces the viewer
            straddles = !inside && !offscreen;
        }
        if (go) {
            // T_normal_z_negative, TransLn.cpp:218-223
            const long long c = (long long)wsub(t[1].v[0], t[0].v[0]) * (long long)wsub(t[2].v[1], t[0].v[1]) -
                                (long long)wsub(t[1].v[1], t[0].v[1]) * (long long)wsub(t[2].v[0], t[0].v[0]);
            if (!(c < 0)) go = false;
        }
        if (go) {
            drawn = true;
            const int miny = min(t[1].v[1], min(t[2].v[1], t[3].v[1])), maxy = max(t[1].v[1], max(t[2].v[1], t[3].v[1]));
            minx = min(t[1].v[0], min(t[2].v[0], t[3].v[0]));
            maxx = max(t[1].v[0], max(t[2].v[0], t[3].v[0]));
            // rows outside the band of this context are another context's to draw
            by0 = max(miny, a.band_y0);
            by1 = min(maxy, a.band_y1);
            visible = by0 < by1 && !offscreen && !straddles;
        }
        straddles = straddles && go;
    }
    // Output slots.  About half of a soup's triangles face away, and writing records with
    // holes between them costs more HBM time than writing twice as many without (partial
    // lines): the polygons of this wave that go on -- visible here, or left to the general
    // kernel -- take the slots base .. base+count-1 of the wave's 64 in lane order, and the
    // rest of the 64 headers is zeroed.  Slot order is polygon order, and everything after
    // set-up (binning, fill, draw-order ties) knows a polygon by its slot.
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long keep = __ballot(visible || slow || straddles);
    const int base = poly & ~63, count = __popcll(keep);
    const int slot = base + __popcll(keep & below);
    // A lane's record is 16*RU contiguous bytes, RU pieces apart from its neighbour's: stored
    // lane by lane every store instruction would touch 64 different lines.  The wave builds
    // its records in LDS, 32 slots at a time, and writes the run of slots it owns with
    // consecutive 16-byte stores.  Slots left to the queue kernels get zeros here and their
    // records later.
    int4 R[RU];
#pragma unroll
    for (int q = 0; q < RU; q++) R[q] = make_int4(0, 0, 0, 0);
    if (visible) {
        int col = colour;
        if (kind == KIND_FLAT) {
            const int4 ax = a.polyaux ? ((const int4 *)a.polyaux)[poly] : make_int4(0, 0, 0, 0);
            col = cl_light(colour, ax.x, ax.y, ax.z);
        }
        R[0] = make_int4(by0 | (by1 << 16), minx | (maxx << 16), col, 3 | (kind << 8) | ((tex < 0 ? 0 : tex) << 16));
#pragma unroll
        for (int e = 0; e < 3; e++) {
            int rec[ESZ];
            make_edge_record<DIM>(t[1 + e], t[e == 2 ? 1 : 2 + e], rec);
            R[1 + e] = make_int4(rec[0], rec[1], rec[2], rec[3]);
#pragma unroll
            for (int q = 1; q < ESZ / 4; q++)
                R[4 + e * (ESZ / 4 - 1) + (q - 1)] = make_int4(rec[4 * q], rec[4 * q + 1], rec[4 * q + 2], rec[4 * q + 3]);
        }
        ((int2 *)a.hdr)[slot] = make_int2(R[0].x, R[0].y);
        if (a.tile_count) count_tiles(a, by0, by1, minx, maxx);
    }
    {
        const bool kept = (keep >> lane) & 1ull;
        const int rank = slot - base;
        int4 *out = (int4 *)a.edges + (size_t)base * RU;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            if (count > 32 * half) {
                if (kept && (rank >> 5) == half) {
#pragma unroll
                    for (int q = 0; q < RU; q++) stg[(rank & 31) * RU + q] = R[q];
                }
                LDS_WAVE_SYNC();
                const int n = (min(count, 32 * half + 32) - 32 * half) * RU;
                for (int u = lane; u < n; u += 64) out[32 * half * RU + u] = stg[u];
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    if (lane >= count && base + lane < a.np) ((int2 *)a.hdr)[base + lane] = make_int2(0, 0);
    // queue the rest for the general kernel as {polygon, slot}: one atomic per wave,
    // spread over QUEUES sub-queues by workgroup index (thousands of waves adding to one
    // word would queue behind each other); sub-queue q holds at most `queue_cap` entries
    // by construction
    const unsigned long long sm = __ballot(slow);
    if (sm) {
        const int leader = __ffsll((long long)sm) - 1;
        const int q = blockIdx.x % QUEUES;
        unsigned qbase = 0;
        if (lane == leader) qbase = atomicAdd(a.queue_count + q * QUEUE_COUNT_STRIDE, (unsigned)__popcll(sm));
        qbase = __shfl(qbase, leader);
        if (slow) a.queue[(size_t)q * a.queue_cap + qbase + __popcll(sm & below)] = make_uint2((unsigned)poly, (unsigned)slot);
    }
    // Triangles that meet a screen border: their four projected tuples go to the screen
    // queue, as {first, edges | kind<<8, colour, tex} + tuples, for the set-up kernel in
    // screen mode (what M_render_polygon hands to G_*_polygon, EngBase.cpp:451-545).
    const unsigned long long qm = __ballot(straddles);
    if (qm) {
        const int leader = __ffsll((long long)qm) - 1;
        const int q = blockIdx.x % QUEUES;
        unsigned qbase = 0;
        if (lane == leader) qbase = atomicAdd(a.sq_count + q * QUEUE_COUNT_STRIDE, (unsigned)__popcll(qm));
        qbase = __shfl(qbase, leader);
        if (straddles) {
            const unsigned idx = (unsigned)q * a.queue_cap + qbase + __popcll(qm & below);
            int col = colour;
            if (kind == KIND_FLAT) {
                const int4 ax = a.polyaux ? ((const int4 *)a.polyaux)[poly] : make_int4(0, 0, 0, 0);
                col = cl_light(colour, ax.x, ax.y, ax.z);
            }
            ((int4 *)a.sq_desc)[idx] = make_int4((int)(idx * 4u), 3 | (kind << 8), col, tex < 0 ? 0 : tex);
            int *dst = a.sq_tuples + (size_t)idx * 4 * DIM;
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < DIM; j++) dst[i * DIM + j] = t[i].v[j];
            a.sq_queue[idx] = make_uint2(idx, (unsigned)slot);
        }
    }
    Status *st = a.status + (blockIdx.x % STATUS_SLOTS);
    const unsigned long long dm = __ballot(drawn);
    if (dm && (threadIdx.x & 63) == (__ffsll((long long)dm) - 1)) atomicAdd(&st->polys_drawn, (unsigned long long)__popcll(dm));
    if (DIM == 8 && a.detect_order) {
        const unsigned long long om = __ballot(visible && vary);
        if (om && (threadIdx.x & 63) == (__ffsll((long long)om) - 1)) *a.order_flag = a.order_gen;
    }
}


// One polygon per lane.  The stages of M_render_polygon (EngBase.cpp:379-552) and of
// G_*_polygon up to the edge scan (GraphicsPoly.cpp:318-323) are chained as a stream:
// a stage hands each tuple it produces to the next stage at once, so a lane holds a
// few tuples in registers and no per-lane arrays exist.  Each stage of the reference
// is a function of the tuple sequence of the stage before it, so chaining them gives
// the same tuples in the same order.
//
//   Z stage  C_polygon_z_clipping  edges (t[i-1], t[i])          -> z-clipped tuples
//   P        T_perspective on each of them; the first three feed T_normal_z_negative
//   X stage  C_polygon_x_clipping  edges of the projected polygon -> x-clipped tuples
//   S stage  GI_scan front end     y clip of each edge, increments -> edge records
//
// Edge records go to the record and planes of the polygon's output slot (`slot`: given by the fast
// kernel for queued polygons, the polygon's own index otherwise) while the stream runs
// and are made visible by the header written at the end (n_edges == 0: nothing to draw).
//
// SCREEN = true starts at the X stage with already-projected tuples (the G_*_polygon
// entry points); polys[] then holds {first, length, colour, tex} descriptors.
// `first_known` (screen mode, >= 0): the polygon's first tuple, where the caller can tell it
// without the descriptor, so that descriptor and tuples are fetched together.
template <int DIM, bool SCREEN, bool MULTI = false>
__device__ __forceinline__ void setup_one(const SetupArgs &a, const int poly, const int slot, const int first_known = -1)
{
    constexpr int NA = DIM - 2;
    constexpr int ESZ = edge_ints(NA);
    struct { int yy, xx, colour, meta; } hdr;
    hdr.yy = 0; hdr.xx = 0; hdr.colour = 0; hdr.meta = 0;

    const int4 pd = ((const int4 *)a.polys)[poly];
    int colour, tex, e, first;
    if (SCREEN) { first = first_known >= 0 ? first_known : pd.x; e = pd.y & 0xff; colour = pd.z; tex = pd.w; }
    else        { colour = pd.x; tex = pd.y; e = pd.z; first = pd.w; }

    bool bad = false;        // polygon cannot be processed (limits); counted, not drawn
    bool unsupported = false;
    int kind = a.render_type;
    if (!SCREEN && kind == KIND_TEX_LIN && tex < 0) kind = KIND_SHADED;   // EngBase.cpp:470,518-522
    if (SCREEN && (pd.y >> 8) != 0) kind = pd.y >> 8;      // screen queue: the fast kernel has chosen the fill

    bool go = true;
    if (e < 2 || e + 1 > 16 || e > a.max_edges || first < 0 || (long long)first + e + 1 > (long long)a.npv) {
        bad = true; go = false;
    }
    VertexSource<MULTI && !SCREEN> vs;
    if (!SCREEN && !vs.init(a, poly)) { bad = true; go = false; }
    if (kind >= KIND_TEX_LIN && (tex < 0 || tex >= 64)) { unsupported = true; go = false; }
    if (SCREEN && kind == KIND_TEX_PRP && !a.prp_given) { unsupported = true; go = false; }

    // ---- read the polygon's tuples.  Polygons of up to PRE tuples (triangles, quads) are
    // read up front -- all records at once, then all vertices at once -- so that a lane
    // pays three dependent memory round trips (descriptor, records, vertices) instead of
    // two per tuple; longer polygons read each tuple when the stream needs it.
    constexpr int PRE = 5;
    Tup<DIM> tin[PRE];
#pragma unroll
    for (int i = 0; i < PRE; i++) tin[i] = Tup<DIM>();
    const bool pre = go && e + 1 <= PRE;
    if (pre) {
        if (SCREEN) {
#pragma unroll
            for (int i = 0; i < PRE; i++)
                if (i <= e) {
                    const int *src = a.polyverts + (size_t)(first + i) * DIM;
#pragma unroll
                    for (int j = 0; j < DIM; j++) tin[i].v[j] = src[j];
                }
        } else {
            int2 ra[PRE], rb[PRE], rc[PRE];
#pragma unroll
            for (int i = 0; i < PRE; i++) {
                ra[i] = make_int2(0, 0); rb[i] = ra[i]; rc[i] = ra[i];
                if (i <= e) {
                    const int2 *rec = (const int2 *)(a.polyverts + (size_t)(first + i) * 6);
                    ra[i] = rec[0];
                    if (DIM >= 6) rb[i] = rec[1];
                    if (DIM == 8) rc[i] = rec[2];
                }
            }
#pragma unroll
            for (int i = 0; i < PRE; i++)
                if (i <= e) {
                    if (!vs.valid(ra[i].x)) { bad = true; go = false; }
                    else {
                        // CI_construct_tuples, EngBase.cpp:58-137
                        int4 p = vs.load(a, ra[i].x);
                        tin[i].v[0] = p.x; tin[i].v[1] = p.y; tin[i].v[2] = p.z;
                        if (DIM >= 6) { tin[i].v[3] = wshl(ra[i].y, 16); tin[i].v[4] = wshl(rb[i].x, 16); tin[i].v[5] = wshl(rb[i].y, 16); }
                        if (DIM == 8) {
                            bool tx = kind == KIND_TEX_LIN;
                            tin[i].v[DIM - 2] = tx ? wshl(rc[i].x, 16) : 0;
                            tin[i].v[DIM - 1] = tx ? wshl(rc[i].y, 16) : 0;
                        }
                    }
                }
        }
    }

    // ---- pass 1 (object mode): extent of the e+1 tuples, C_volume_clipping (Clipp3D.cpp:46-58)
    bool zclip = false;
    if (!SCREEN && go) {
        int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
        if (pre) {
#pragma unroll
            for (int i = 0; i < PRE; i++)
                if (i <= e) {
#pragma unroll
                    for (int j = 0; j < 3; j++) { mn[j] = min(mn[j], tin[i].v[j]); mx[j] = max(mx[j], tin[i].v[j]); }
                }
        } else {
            for (int i = 0; i <= e; i++) {
                int idx = a.polyverts[(size_t)(first + i) * 6];
                if (!vs.valid(idx)) { bad = true; go = false; break; }
                int4 p = vs.load(a, idx);
                mn[0] = min(mn[0], p.x); mx[0] = max(mx[0], p.x);
                mn[1] = min(mn[1], p.y); mx[1] = max(mx[1], p.y);
                mn[2] = min(mn[2], p.z); mx[2] = max(mx[2], p.z);
            }
        }
        if (go) {
            if ((mx[2] < mn[0]) || (mx[2] < mn[1]) || (mx[2] < -mx[0]) || (mx[2] < -mx[1]) || (mx[2] <= 10))
                go = false;
            else
                zclip = mn[2] < 10;
        }
    }

    // ---- the stream
    int ne = 0;                       // edge records written
    int miny = INT_MAX, maxy = INT_MIN, minx = INT_MAX, maxx = INT_MIN;
    Tup<DIM> zprev, zfirst;           // Z stage: previous input tuple / first output tuple
    Tup<DIM> xprev, sprev, sfirst;    // X stage: previous input; S stage: previous input, first
    zprev = Tup<DIM>(); zfirst = zprev; xprev = zprev; sprev = zprev; sfirst = zprev;
    int zn = 0, xn = 0, sn = 0;       // tuples emitted by the Z stage / taken by X / taken by S
    int px[3] = {0, 0, 0}, py[3] = {0, 0, 0};
    bool lights_vary = false;         // textured: vertex lights differ (the fill is then order dependent)
    int l0r = 0, l0g = 0, l0b = 0;
    int first_view_z = 0;             // clipped[2] of EngBase.cpp:492
    Tup<DIM> org0 = zprev, org1 = zprev, org2 = zprev;   // original1[0..2] of EngBase.cpp:495-498 (textured only)
    Tup<DIM> zfirst_view = zprev;                         // `clipped` tuple 0, view space
    const int xc = a.W / 2, yc = a.H / 2;

    // object mode: iterations 0..e read the tuples, iteration e+1 is the closing event
    // of a z-clipped polygon.  screen mode: iterations 0..e read the tuples.
    const int iters = SCREEN ? e + 1 : e + 2;
    for (int it = 0; go && it < iters; it++) {
        Tup<DIM> zo0 = zprev, zo1 = zprev;      // up to two tuples leave the Z stage per step
        bool zclosing0 = false;
        int cz = 0;

        if (it <= e) {
            Tup<DIM> t;
            if (pre) {
                t = tin[0];
#pragma unroll
                for (int i = 1; i < PRE; i++) t = sel(it == i, tin[i], t);
            } else if (SCREEN) {
                const int *src = a.polyverts + (size_t)(first + it) * DIM;
#pragma unroll
                for (int j = 0; j < DIM; j++) t.v[j] = src[j];
            } else {
                // CI_construct_tuples, EngBase.cpp:58-137
                const int *rec = a.polyverts + (size_t)(first + it) * 6;
                int4 p = vs.load(a, rec[0]);
                t.v[0] = p.x; t.v[1] = p.y; t.v[2] = p.z;
                if (DIM >= 6) { t.v[3] = wshl(rec[1], 16); t.v[4] = wshl(rec[2], 16); t.v[5] = wshl(rec[3], 16); }
                if (DIM == 8) {
                    bool tx = kind == KIND_TEX_LIN;
                    t.v[DIM - 2] = tx ? wshl(rec[4], 16) : 0;
                    t.v[DIM - 1] = tx ? wshl(rec[5], 16) : 0;
                }
            }
            if (DIM == 8) {
                if (it == 0) { l0r = t.v[3]; l0g = t.v[4]; l0b = t.v[5]; }
                else lights_vary = lights_vary || t.v[3] != l0r || t.v[4] != l0g || t.v[5] != l0b;
            }
            if (DIM == 8 && !SCREEN) {
                org0 = sel(it == 0, t, org0);
                org1 = sel(it == 1, t, org1);
                org2 = sel(it == 2, t, org2);
            }
            if (SCREEN) {
                zo0 = t; zclosing0 = (it == e); cz = 1;
            } else {
                if (!zclip) {
                    zo0 = t; zclosing0 = (it == e); cz = 1;
                } else if (it > 0) {
                    // C_polygon_z_clipping, Clipp3D.cpp:137-155
                    Tup<DIM> o1, o2;
                    bool moved;
                    if (clip_line<DIM, 2, false>(zprev, t, 10, 0, o1, o2, moved, bad)) {
                        zo0 = sel(moved, o1, o2);
                        zo1 = o2;
                        cz = moved ? 2 : 1;
                    }
                }
                zprev = t;
            }
        } else if (zclip) {
            // closing tuple of the z-clipped polygon = its first tuple (Clipp3D.cpp:156)
            if (zn > 0) { zo0 = zfirst; zclosing0 = true; cz = 1; }
        }

        for (int k = 0; k < cz; k++) {
            Tup<DIM> p = sel(k != 0, zo1, zo0);
            bool closing = k == 0 && zclosing0;
            if (!SCREEN) {
                if (!closing) {
                    if (zn == 0) { zfirst = p; first_view_z = p.v[2]; if (DIM == 8) zfirst_view = p; }
                    zn++;
                    if (zn + 1 > 16) bad = true;
                }
                // T_perspective, TransBs.cpp:425-432
                const int z = p.v[2];               // >= 10 after the volume test and the near clip
                const double rz = drcp(z);
                p.v[0] = wadd(persp_div(p.v[0], a.log_focus, z, rz), xc);
                p.v[1] = wadd(persp_div(p.v[1], a.log_focus, z, rz), yc);
                p.v[2] = tdivr(-0x7ffff, rz);
                int seq = closing ? zn : zn - 1;
                if (seq < 3) {
                    // static indices keep px/py in registers
                    if (seq == 0) { px[0] = p.v[0]; py[0] = p.v[1]; }
                    else if (seq == 1) { px[1] = p.v[0]; py[1] = p.v[1]; }
                    else { px[2] = p.v[0]; py[2] = p.v[1]; }
                }
            }
            // ---- X stage: C_polygon_x_clipping, Clipp2D.cpp:253-272
            Tup<DIM> xo0 = p, xo1 = p;
            int cx = 0;
            if (xn > 0) {
                Tup<DIM> o1, o2;
                bool moved;
                if (clip_line<DIM, 0, true>(xprev, p, a.cx0, a.cx1, o1, o2, moved, bad)) {
                    xo0 = sel(moved, o1, o2);
                    xo1 = o2;
                    cx = moved ? 2 : 1;
                }
            }
            xprev = p;
            xn++;
            // ---- S stage: edges (s[i], s[i+1]) of the x-clipped polygon, closing edge
            // back to its first tuple (Clipp2D.cpp:272), each through GI_scan's front end
            const int cs = cx + (closing ? 1 : 0);
            for (int j = 0; j < cs; j++) {
                bool sclosing = j >= cx;
                Tup<DIM> s = sel(sclosing, sfirst, sel(j != 0, xo1, xo0));
                if (sclosing && sn == 0) break;
                if (!sclosing) {
                    minx = min(minx, s.v[0]); maxx = max(maxx, s.v[0]);
                }
                if (sn > 0) {
                    // GI_scan, GraphicsPoly.cpp:134-176
                    Tup<DIM> v1, v2;
                    bool moved;
                    if (clip_line<DIM, 1, true>(sprev, s, a.cy0, a.cy1, v1, v2, moved, bad)) {
                        miny = min(miny, min(v1.v[1], v2.v[1]));
                        maxy = max(maxy, max(v1.v[1], v2.v[1]));
                        if (ne < a.emax) store_edge_record<DIM>(a, slot, ne, v1, v2);
                        ne++;
                    }
                }
                if (!sclosing) {
                    if (sn == 0) sfirst = s;
                    sn++;
                    if (sn + 1 > 16) bad = true;
                }
                sprev = s;
            }
        }
        if (bad) go = false;
    }

    bool draw = go && !bad && !unsupported;
    bool mapper_failed = false;      // faced the viewer, but the perspective mapper cannot be set up
    if (draw && !SCREEN) {
        if (zn < 2) draw = false;
        else {
            // T_normal_z_negative, TransLn.cpp:218-223, on projected tuples 0,1,2
            long long c = (long long)wsub(px[1], px[0]) * (long long)wsub(py[2], py[0]) -
                          (long long)wsub(py[1], py[0]) * (long long)wsub(px[2], px[0]);
            if (!(c < 0)) draw = false;
        }
        if (DIM == 8 && draw && kind == KIND_TEX_LIN && (a.force_linear || first_view_z < 1024)) {
            // EngBase.cpp:491-506 routes this polygon to the perspective mapper
            const int4 ax = a.polyaux ? ((const int4 *)a.polyaux)[poly] : make_int4(0, 0, 0, 0);
            const int log_space = ax.w >> 8;
            int u[3], v[3];
            Tup<8> q0, q1, q2, O;
#pragma unroll
            for (int j = 0; j < 8; j++) { q0.v[j] = org0.v[j % DIM]; q1.v[j] = org1.v[j % DIM]; q2.v[j] = org2.v[j % DIM]; O.v[j] = zfirst_view.v[j % DIM]; }
            if (!a.prp || log_space < 0 || log_space > 24 || !tmapping_vectors(ax.w & 0x0c, q0, q1, q2, log_space, u, v)) {
                unsupported = true; draw = false; mapper_failed = true;
            } else {
                kind = KIND_TEX_PRP;
                // GraphicsPoly.cpp:563-578
                const int x0 = O.v[0], y0 = O.v[1], z0 = O.v[2];
                int *pr = a.prp + (size_t)slot * PRP_INTS;
                int4 c0, c1, c2;
                c0.x = wsub(wmul(v[2], y0), wmul(v[1], z0));      // ui
                c0.y = wsub(wmul(v[0], z0), wmul(v[2], x0));      // uj
                c0.z = wsub(wmul(v[1], x0), wmul(v[0], y0));      // uc
                c0.w = wsub(wmul(u[1], z0), wmul(u[2], y0));      // vi
                c1.x = wsub(wmul(u[2], x0), wmul(u[0], z0));      // vj
                c1.y = wsub(wmul(u[0], y0), wmul(u[1], x0));      // vc
                c1.z = wsub(wmul(v[1], u[2]), wmul(v[2], u[1]));  // zi
                c1.w = wsub(wmul(v[2], u[0]), wmul(v[0], u[2]));  // zj
                c2.x = wsub(wmul(v[0], u[1]), wmul(v[1], u[0]));  // zc
                c2.y = O.v[6]; c2.z = O.v[7]; c2.w = log_space;
                ((int4 *)pr)[0] = c0; ((int4 *)pr)[1] = c1; ((int4 *)pr)[2] = c2;
            }
        }
    }
    if (ne > a.emax) { bad = true; draw = false; }

    bool reached_fill = draw || mapper_failed;      // M_render_polygon got as far as choosing the fill
    if (draw && !(miny < maxy && ne > 0)) draw = false;
    if (draw) {
        // rows outside the band of this context are another context's to draw
        miny = max(miny, a.band_y0);
        maxy = min(maxy, a.band_y1);
        if (!(miny < maxy)) draw = false;
    }

    if (draw) {
        int col = colour;
        if (kind == KIND_FLAT && !SCREEN) {
            int4 ax = a.polyaux ? ((const int4 *)a.polyaux)[poly] : make_int4(0, 0, 0, 0);
            col = cl_light(colour, ax.x, ax.y, ax.z);
        }
        hdr.yy = miny | (maxy << 16);
        hdr.xx = minx | (maxx << 16);
        hdr.colour = col;
        hdr.meta = ne | (kind << 8) | ((tex < 0 ? 0 : tex) << 16);
    }
    ((int2 *)a.hdr)[slot] = make_int2(hdr.yy, hdr.xx);
    *record_header(a.edges, ESZ / 4, slot) = make_int4(hdr.yy, hdr.xx, hdr.colour, hdr.meta);
    if (a.tile_count && draw) count_tiles(a, miny, maxy, minx, maxx);
    if (DIM == 8 && a.detect_order) {
        const bool ord = draw && lights_vary && kind >= KIND_TEX_LIN;
        const unsigned long long om = __ballot(ord);
        if (om && (threadIdx.x & 63) == (__ffsll((long long)om) - 1)) *a.order_flag = a.order_gen;
    }

    // counters: one atomic per wave, spread over the status slots
    Status *st = a.status + ((blockIdx.x + blockIdx.y) % STATUS_SLOTS);
    unsigned long long m;
    m = __ballot(reached_fill && a.count_drawn);
    if (m && (threadIdx.x & 63) == (__ffsll((long long)m) - 1)) atomicAdd(&st->polys_drawn, (unsigned long long)__popcll(m));
    m = __ballot(bad);
    if (m && (threadIdx.x & 63) == (__ffsll((long long)m) - 1)) atomicAdd(&st->overflow_polys, (unsigned long long)__popcll(m));
    m = __ballot(unsupported);
    if (m && (threadIdx.x & 63) == (__ffsll((long long)m) - 1)) atomicAdd(&st->unsupported_polys, (unsigned long long)__popcll(m));
}

template <int DIM, bool SCREEN>
__global__ void __launch_bounds__(SETUP_THREADS) setup_kernel(SetupArgs a)
{
    __builtin_amdgcn_s_setprio(3);          // see setup_fast_kernel
    if (a.queue) {
        // second stage: only the polygons that the fast kernel passed on.  Sub-queue
        // blockIdx.y is shared by the few workgroups of row y of the grid, which stride
        // through it; its length is only known on the device.
        const unsigned cnt = min(a.queue_count[blockIdx.y * QUEUE_COUNT_STRIDE], a.queue_cap);
        for (unsigned qi = blockIdx.x * blockDim.x + threadIdx.x; qi < cnt; qi += gridDim.x * blockDim.x) {
            const uint2 e = a.queue[(size_t)blockIdx.y * a.queue_cap + qi];
            if (e.x < a.queue_index_limit && e.y < (unsigned)a.np) setup_one<DIM, SCREEN>(a, (int)e.x, (int)e.y);
        }
    } else {
        const int poly = blockIdx.x * blockDim.x + threadIdx.x;
        if (poly < a.np) setup_one<DIM, SCREEN>(a, poly, poly);
    }
}

// Both queues of the fast kernel in one launch: rows 0..QUEUES-1 of the grid take the
// projected triangles (screen mode), rows QUEUES..2*QUEUES-1 the polygons that need the
// whole path.  One launch instead of two: the second kernel is nearly always almost empty,
// and an empty launch still costs its dispatch, several times over when the chip is busy
// with the fill of the previous batch.
template <int DIM, bool MULTI>
__device__ __forceinline__ void setup_queues(const SetupArgs &a)
{
    __builtin_amdgcn_s_setprio(3);          // see setup_fast_kernel
    if (blockIdx.y < QUEUES) {
        SetupArgs q = a;
        q.polys = a.sq_desc; q.polyverts = a.sq_tuples; q.polyaux = nullptr;
        q.npv = (int)min((unsigned long long)QUEUES * a.queue_cap * 4ull, 0x7fffffffull);
        q.queue_index_limit = QUEUES * a.queue_cap;
        q.count_drawn = 0;              // the fast kernel has counted them
        q.use_matrix = 0;
        const unsigned cnt = min(a.sq_count[blockIdx.y * QUEUE_COUNT_STRIDE], a.queue_cap);
        for (unsigned qi = blockIdx.x * blockDim.x + threadIdx.x; qi < cnt; qi += gridDim.x * blockDim.x) {
            // entry i of the screen queue describes itself (the fast kernel wrote {i, slot}, descriptor i
            // and tuples 4i ..): the slot, the descriptor and the tuples are fetched together
            const unsigned idx = blockIdx.y * a.queue_cap + qi;
            const uint2 e = a.sq_queue[idx];
            if (idx < q.queue_index_limit && e.y < (unsigned)a.np) setup_one<DIM, true>(q, (int)idx, (int)e.y, (int)(idx * 4u));
        }
    } else {
        const unsigned y = blockIdx.y - QUEUES;
        const unsigned cnt = min(a.queue_count[y * QUEUE_COUNT_STRIDE], a.queue_cap);
        for (unsigned qi = blockIdx.x * blockDim.x + threadIdx.x; qi < cnt; qi += gridDim.x * blockDim.x) {
            const uint2 e = a.queue[(size_t)y * a.queue_cap + qi];
            if (e.x < a.queue_index_limit && e.y < (unsigned)a.np) setup_one<DIM, false, MULTI>(a, (int)e.x, (int)e.y);
        }
    }
}
// One kernel per tuple size, because their register budgets differ.  While fills run, three or
// four of their waves sit on every SIMD (86 to 106 VGPRs each) and a wave of this kernel has to
// find room beside them: the flat and Gouraud sizes are held to a third of the register file
// (the Gouraud one spills 21 registers for it; C2-small with two frames in flight 0.1243 ->
// 0.1201 ms per frame).  The textured size needs all 256: held to 168 it spills 257 and C3 at 4K
// goes from 1.13 to 1.21 ms.
template <int DIM> __global__ void setup_queues_kernel(SetupArgs a);
template <> __global__ void __launch_bounds__(SETUP_THREADS) __attribute__((amdgpu_waves_per_eu(3))) setup_queues_kernel<3>(SetupArgs a) { setup_queues<3, false>(a); }
template <> __global__ void __launch_bounds__(SETUP_THREADS) __attribute__((amdgpu_waves_per_eu(3))) setup_queues_kernel<6>(SetupArgs a) { setup_queues<6, false>(a); }
template <> __global__ void __launch_bounds__(SETUP_THREADS, 2) setup_queues_kernel<8>(SetupArgs a) { setup_queues<8, false>(a); }
// the same for a batch that draws a table of objects (the polygon's object gives matrix and vertex range)
template <int DIM> __global__ void setup_queues_objects_kernel(SetupArgs a);
template <> __global__ void __launch_bounds__(SETUP_THREADS) __attribute__((amdgpu_waves_per_eu(3))) setup_queues_objects_kernel<3>(SetupArgs a) { setup_queues<3, true>(a); }
template <> __global__ void __launch_bounds__(SETUP_THREADS) __attribute__((amdgpu_waves_per_eu(3))) setup_queues_objects_kernel<6>(SetupArgs a) { setup_queues<6, true>(a); }
template <> __global__ void __launch_bounds__(SETUP_THREADS, 2) setup_queues_objects_kernel<8>(SetupArgs a) { setup_queues<8, true>(a); }

// ------------------------------------------------------------------ binning

// Tile lists are built by a counting sort over (polygon, tile) references: count per
// tile, exclusive scan, scatter.  A workgroup takes a contiguous chunk of polygons and
// keeps its own per-tile histogram in LDS, so the chip-wide atomics shrink from one per
// reference to one per (chunk, touched tile).  The order of a list's entries does not
// matter: the fill resolves by (depth, draw index).
// PER polygons a thread: 8 (chunks of 4096), and 16 for batches of four million polygons and more, where a
// workgroup's flush of its histogram -- one chip-wide add per touched tile -- is what the kernels last
// (ten million triangles at 4K: 1.07 -> 1.05 ms per frame; one million at 1080p lose 2 % that way).
constexpr int BIN_THREADS = 512;
constexpr int BIN_LARGE_BATCH = 1 << 22;

__device__ __forceinline__ bool prim_tiles(const int2 h, int twl, int &tx0, int &tx1, int &ty0, int &ty1)
{
    if (h.x == 0) return false;          // no rows: nothing to draw
    int miny = h.x & 0xffff, maxy = (h.x >> 16) & 0xffff, minx = h.y & 0xffff, maxx = (h.y >> 16) & 0xffff;
    tx0 = minx >> twl; tx1 = maxx >> twl; ty0 = miny / TH; ty1 = (maxy - 1) / TH;
    return true;
}

// (`order_flag` non-null: a textured batch.  If its set-up has found that it must be filled in
// draw order -- or the caller has asked for that, `order_force` -- the tile lists have no reader:
// the tile fill returns at once and the draw-order kernels work from the headers.)
template <int BIN_PER_THREAD>
__global__ void __launch_bounds__(BIN_THREADS) count_kernel(const PrimHdr *__restrict__ hdr, int np, int twl, int tiles_x,
                                                            int ntiles, unsigned *__restrict__ count,
                                                            const int *__restrict__ order_flag, int order_gen, int order_force)
{
    __builtin_amdgcn_s_setprio(3);          // see setup_fast_kernel
    extern __shared__ unsigned hist[];          // ntiles words (the launch sizes it: a small request finds room beside running fills)
    if (order_flag && (order_force || *order_flag == order_gen)) return;
    for (int i = threadIdx.x; i < ntiles; i += BIN_THREADS) hist[i] = 0;
    __syncthreads();
    constexpr int BIN_CHUNK = BIN_THREADS * BIN_PER_THREAD;
    const int base = blockIdx.x * BIN_CHUNK;
#pragma unroll
    for (int k = 0; k < BIN_PER_THREAD; k++) {
        int poly = base + k * BIN_THREADS + threadIdx.x;
        if (poly < np) {
            int tx0, tx1, ty0, ty1;
            if (prim_tiles(((const int2 *)hdr)[poly], twl, tx0, tx1, ty0, ty1))
                for (int ty = ty0; ty <= ty1; ty++)
                    for (int tx = tx0; tx <= tx1; tx++) atomicAdd(&hist[ty * tiles_x + tx], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ntiles; i += BIN_THREADS)
        if (hist[i]) atomicAdd(&count[i], hist[i]);
}

template <int BIN_PER_THREAD>
__global__ void __launch_bounds__(BIN_THREADS) scatter_kernel(const PrimHdr *__restrict__ hdr, int np, int twl, int tiles_x,
                                                              int ntiles, const unsigned *__restrict__ offset,
                                                              unsigned *__restrict__ cursor, unsigned *__restrict__ list,
                                                              unsigned long long capacity,
                                                              const int *__restrict__ order_flag, int order_gen, int order_force)
{
    __builtin_amdgcn_s_setprio(3);          // see setup_fast_kernel
    extern __shared__ unsigned hist[];            // ntiles words: count, then the chunk's base slot inside the tile's list
    if (order_flag && (order_force || *order_flag == order_gen)) return;
    for (int i = threadIdx.x; i < ntiles; i += BIN_THREADS) hist[i] = 0;
    __syncthreads();
    constexpr int BIN_CHUNK = BIN_THREADS * BIN_PER_THREAD;
    const int base = blockIdx.x * BIN_CHUNK;
    int2 hs[BIN_PER_THREAD];
#pragma unroll
    for (int k = 0; k < BIN_PER_THREAD; k++) {
        int poly = base + k * BIN_THREADS + threadIdx.x;
        hs[k] = poly < np ? ((const int2 *)hdr)[poly] : make_int2(0, 0);
        int tx0, tx1, ty0, ty1;
        if (prim_tiles(hs[k], twl, tx0, tx1, ty0, ty1))
            for (int ty = ty0; ty <= ty1; ty++)
                for (int tx = tx0; tx <= tx1; tx++) atomicAdd(&hist[ty * tiles_x + tx], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ntiles; i += BIN_THREADS) {
        unsigned h = hist[i];
        if (h) hist[i] = offset[i] + atomicAdd(&cursor[i], h);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BIN_PER_THREAD; k++) {
        int poly = base + k * BIN_THREADS + threadIdx.x;
        int tx0, tx1, ty0, ty1;
        if (prim_tiles(hs[k], twl, tx0, tx1, ty0, ty1))
            for (int ty = ty0; ty <= ty1; ty++)
                for (int tx = tx0; tx <= tx1; tx++) {
                    unsigned slot = atomicAdd(&hist[ty * tiles_x + tx], 1u);
                    if ((unsigned long long)slot < capacity) list[slot] = (unsigned)poly;
                }
    }
}

// direct versions for screens with more tiles than the LDS histogram holds
__global__ void __launch_bounds__(256) count_direct_kernel(const PrimHdr *__restrict__ hdr, int np, int twl, int tiles_x,
                                                           unsigned *__restrict__ count)
{
    int poly = blockIdx.x * blockDim.x + threadIdx.x;
    if (poly >= np) return;
    int tx0, tx1, ty0, ty1;
    if (prim_tiles(((const int2 *)hdr)[poly], twl, tx0, tx1, ty0, ty1))
        for (int ty = ty0; ty <= ty1; ty++)
            for (int tx = tx0; tx <= tx1; tx++) atomicAdd(&count[ty * tiles_x + tx], 1u);
}

__global__ void __launch_bounds__(256) scatter_direct_kernel(const PrimHdr *__restrict__ hdr, int np, int twl, int tiles_x,
                                                             const unsigned *__restrict__ offset,
                                                             unsigned *__restrict__ cursor, unsigned *__restrict__ list,
                                                             unsigned long long capacity)
{
    int poly = blockIdx.x * blockDim.x + threadIdx.x;
    if (poly >= np) return;
    int tx0, tx1, ty0, ty1;
    if (prim_tiles(((const int2 *)hdr)[poly], twl, tx0, tx1, ty0, ty1))
        for (int ty = ty0; ty <= ty1; ty++)
            for (int tx = tx0; tx <= tx1; tx++) {
                int tile = ty * tiles_x + tx;
                unsigned slot = offset[tile] + atomicAdd(&cursor[tile], 1u);
                if ((unsigned long long)slot < capacity) list[slot] = (unsigned)poly;
            }
}

// Exclusive prefix sum of the per-tile reference counts, one workgroup of 1024 threads.
// Tiles are taken 1024 at a time, coalesced.  Pass 1 leaves every wave's sum of every
// step in LDS (no barrier inside the loop, so the loads of all steps are in flight
// together); the wave sums are scanned; pass 2 reads the counts again (from L2) and writes
// the offsets.  Also zeroes the scatter cursors, and leaves the counts and the set-up
// sub-queue counters at zero for the next batch that uses these arrays (they are zero
// when allocated), so no batch needs a memset.
constexpr int SCAN_MAX_STEPS = 128;        // 131072 tiles: the largest screen (16384 x 16384)
__global__ void __launch_bounds__(1024) scan_kernel(unsigned *__restrict__ count, unsigned *__restrict__ offset,
                                                    unsigned *__restrict__ cursor, int n,
                                                    unsigned long long capacity, unsigned *__restrict__ queue_count,
                                                    Status *status)
{
    __builtin_amdgcn_s_setprio(3);          // see setup_fast_kernel
    __shared__ unsigned wsum[SCAN_MAX_STEPS * 16];
    __shared__ unsigned tsum[16];
    __shared__ unsigned wmax[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int steps = (n + 1023) / 1024;
    unsigned mine_max = 0;
    for (int st = 0; st < steps; st++) {
        const int i = st * 1024 + t;
        unsigned s = i < n ? count[i] : 0u;
        mine_max = max(mine_max, s);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d);
        if (lane == 0) wsum[st * 16 + w] = s;
    }
    // the longest list of the batch (the host decides from it how many waves the next batches put on a tile)
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mine_max = max(mine_max, (unsigned)__shfl_down((int)mine_max, d));
    if (lane == 0) wmax[w] = mine_max;
    __syncthreads();
    unsigned longest = 0;
    for (int k = 0; k < 16; k++) longest = max(longest, wmax[k]);
    // exclusive scan of the steps*16 wave sums: two per thread
    {
        const int m = steps * 16;
        const unsigned a0 = 2 * t < m ? wsum[2 * t] : 0u, a1 = 2 * t + 1 < m ? wsum[2 * t + 1] : 0u;
        unsigned s = a0 + a1;
        const unsigned mine = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned o = __shfl_up(s, d);
            if (lane >= d) s += o;
        }
        if (lane == 63) tsum[w] = s;
        __syncthreads();
        unsigned pre = 0;
        for (int k = 0; k < w; k++) pre += tsum[k];
        const unsigned ex = pre + s - mine;
        if (2 * t < m) wsum[2 * t] = ex;
        if (2 * t + 1 < m) wsum[2 * t + 1] = ex + a0;
    }
    __syncthreads();
    unsigned total = 0;
    for (int k = 0; k < 16; k++) total += tsum[k];
    for (int st = 0; st < steps; st++) {
        const int i = st * 1024 + t;
        const unsigned v = i < n ? count[i] : 0u;
        unsigned s = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned o = __shfl_up(s, d);
            if (lane >= d) s += o;
        }
        if (i < n) { offset[i] = wsum[st * 16 + w] + s - v; cursor[i] = 0u; count[i] = 0u; }
    }
    if (queue_count && t < 2 * QUEUES) queue_count[t * QUEUE_COUNT_STRIDE] = 0u;      // general queue, then screen queue
    if (t == 0) {
        offset[n] = total;
        atomicAdd(&status->tile_refs, (unsigned long long)total);
        atomicAdd(&status->longest_lists, (unsigned long long)longest);
        if ((unsigned long long)total > capacity)
            atomicAdd(&status->list_overflow, (unsigned long long)total - capacity);
    }
}

// ------------------------------------------------------------------ fill

// The two ends of one polygon row as GI_scan leaves them in G_x_start / G_x_end and
// G_*_start / G_*_end (GraphicsPoly.cpp:178-198), evaluated for one row directly from
// the edge records instead of by walking every edge:
//  - the walk of an edge with long delta L and short delta s has taken
//    h(i) = floor((2*s*i + L) / (2*L)) diagonal steps before point i (d starts at
//    2s-L, +2s each step, -2L more on a diagonal step);
//  - a y-major edge (|dx| <= |dy|) has one point per row, i = |y-y1|, x = x1 +- h(i);
//  - an x-major edge has on row j = |y-y1| the points i_lo(j) .. i_hi(j) with
//    i_lo(j) = ceil((2*L*j - L) / (2*s)) (0 for j = 0), i_hi(j) = i_lo(j+1) - 1
//    (L on the last row);
//  - values at point i are start + i*inc (the additive walk of GraphicsPoly.cpp:196-197
//    in wrapping arithmetic);
//  - the compares of GraphicsPoly.cpp:180,187 are strict, so on equal x the earlier
//    edge keeps the row end.
// edge_row() gives, for one edge and one row, the point index and x of the row's
// leftmost and rightmost point on that edge; it returns false if the edge misses the row.
__device__ __forceinline__ bool edge_row(int xy1, int P, int y, int &ia, int &xa, int &ib, int &xb)
{
    // Branch-free: both the y-major and the x-major forms divide by 2*|dy| (2L for a
    // y-major edge, 2s for an x-major one), so one reciprocal serves both quotients
    // and selects pick the form that applies.  P: the edge's packed deltas (edge_pack).
    const int x1 = xy1 & 0xffff, y1 = (int)((unsigned)xy1 >> 16);
    const int adx = P & 0x7fff, ady = (P >> 15) & 0x7fff;
    const int j = P < 0 ? y1 - y : y - y1;
    const bool hit = (unsigned)j <= (unsigned)ady;
    const bool xmaj = adx > ady;
    const int d2 = ady > 0 ? 2 * ady : 1;                       // a horizontal edge or a point uses no quotient
    const float rd = __builtin_amdgcn_rcpf((float)d2);
    const int a2j = __mul24(2 * adx, j);
    const int nB = a2j - adx + 2 * ady - 1;
    // first quotient: diagonal steps before point j (y-major) or first point of row j
    // (x-major, j >= 1); both are 0 for j = 0.  Second: first point of row j+1 (x-major).
    const int q1 = udiv_r(xmaj ? nB : a2j + ady, d2, rd);
    const int qC = udiv_r(nB + 2 * adx, d2, rd);
    const int olo = j == 0 ? 0 : q1;                            // x distance of the row's first point from x1
    const int ihi = j == ady ? adx : qC - 1;
    const int ohi = xmaj ? ihi : olo;
    const int lo = xmaj ? olo : j, hi = xmaj ? ihi : j;         // point indices of the row's two end points
    const bool fwd = (P & (1 << 30)) != 0;
    ia = fwd ? lo : hi;
    ib = fwd ? hi : lo;
    xa = fwd ? x1 + olo : x1 - ohi;
    xb = fwd ? x1 + ohi : x1 - olo;
    return hit;
}

// The same closed form with everything that does not depend on the row made once per edge:
//   P  = |dx| | |dy| << 15 | (dx >= 0) << 30 | (dy < 0) << 31     (edge_pack: the record's second word)
//   hr = 0.5 / |dy| in double precision                          (1.0 for a horizontal edge)
// Both quotients divide by 2*|dy|; floor(N / (2k)) = floor((N >> 1) / k) for N >= 0, and with
// m = 2*(N >> 1) + 1 the product m * hr lies at least 1/(2k) away from every integer, far more
// than its rounding error (m < 2^31, k < 2^15), so truncating it gives the quotient exactly:
// one 24-bit multiply, two conversions each way and two f64 multiplies per (edge, row)
// instead of a reciprocal and two corrected single-precision quotients.
__device__ __forceinline__ bool edge_row_pre(int xy1, int P, double hr, int y, int &ia, int &xa, int &ib, int &xb)
{
    const int x1 = xy1 & 0xffff, y1 = (int)((unsigned)xy1 >> 16);
    const int adx = P & 0x7fff, ady = (P >> 15) & 0x7fff;
    const int j = P < 0 ? y1 - y : y - y1;
    const bool hit = (unsigned)j <= (unsigned)ady;
    const bool xmaj = adx > ady;
    // numerators halved (see above): y-major  adx*j + (ady >> 1)
    //                                x-major  adx*j + ady + ((-adx - 1) >> 1)   (first point of row j)
    //                                         the same + adx                    (first point of row j + 1)
    const int K2 = 2 * (adx + ady + ((-adx - 1) >> 1)) + 1;
    const int K1 = xmaj ? K2 - 2 * adx : (ady | 1);
    const int t = __mul24(2 * adx, j);
    const int q1 = (int)((double)(t + K1) * hr), qC = (int)((double)(t + K2) * hr);
    const int A = j == 0 ? 0 : q1;                              // x distance of the row's first point from x1
    const int B = j == ady ? adx : qC - 1;
    const int ohi = xmaj ? B : A;
    const int lo = xmaj ? A : j, hi = xmaj ? B : j;             // point indices of the row's two end points
    const bool fwd = (P & (1 << 30)) != 0;
    ia = fwd ? lo : hi;
    ib = fwd ? hi : lo;
    xa = fwd ? x1 + A : x1 - ohi;
    xb = fwd ? x1 + ohi : x1 - A;
    return hit;
}

// Reciprocals of 1..63 as {1/k, 0.5/k}, held in LDS by the tile fill: the divisor of a row's
// increments is its span and the divisor of an edge's point indices is twice its height, both
// short for small polygons; larger ones take drcp().  Entry 0 is {1, 0.5} and never used as a
// divisor.  The values are drcp()'s, so quotients do not depend on which way they were found.
__device__ __forceinline__ void fill_rcp_table(double2 *lut, int t)
{
    if (t < 64) {
        const double r = drcp(t > 0 ? t : 1);
        lut[t] = make_double2(r, 0.5 * r);
    }
}
__device__ __forceinline__ double2 rcp_pair(const double2 *lut, int d)         // d >= 1
{
    double2 l = lut[d < 64 ? d : 63];
    if (__ballot(d >= 64)) {
        int big = d;
        asm volatile("" : "+v"(big));        // keeps the reciprocal inside this branch: as straight-line code it would cost every row more than the table saves
        if (big >= 64) { l.x = drcp(big); l.y = 0.5 * l.x; }
    }
    return l;
}
// 0.5 / d alone (the divisor of an edge's point indices)
__device__ __forceinline__ double rcp_half(const double2 *lut, int d)         // d >= 1
{
    double h = lut[d < 64 ? d : 63].y;
    if (__ballot(d >= 64)) {
        int big = d;
        asm volatile("" : "+v"(big));        // see rcp_pair
        if (big >= 64) h = 0.5 * drcp(big);
    }
    return h;
}
// trunc(n / d) given rcp_pair(d)
__device__ __forceinline__ int tdivp(int n, const double2 rp) { return (int)fma((double)n, rp.x, n < 0 ? -rp.y : rp.y); }

// Row ends with all NA interpolated values (used by the colour pass).
template <int NA>
__device__ __forceinline__ bool row_ends(const int *__restrict__ edges, int np, int poly, int ne, int y, int &beg,
                                         int &end, int (&S)[NA], int (&E)[NA], const int2 *pre = nullptr,
                                         const double2 *lut = nullptr)
{
    constexpr int ESZ = edge_ints(NA);
    int bx = INT_MAX, ex = INT_MIN, be = 0, bi = 0, ee = 0, ei = 0;
    // `pre`: coordinates of edges 0..2 already fetched by the caller; `lut`: the caller's
    // reciprocal table (fill_rcp_table), which selects the form with per-edge constants
    int first = 0;
    if (pre) {
#pragma unroll
        for (int e = 0; e < 3; e++) {
            int ia, xa, ib, xb;
            bool hit;
            if (lut) {
                const int P = pre[e].y;
                const int ady = (P >> 15) & 0x7fff;
                hit = edge_row_pre(pre[e].x, P, rcp_half(lut, ady > 0 ? ady : 1), y, ia, xa, ib, xb);
            } else {
                hit = edge_row(pre[e].x, pre[e].y, y, ia, xa, ib, xb);
            }
            if (e < ne && hit) {
                if (xa < bx) { bx = xa; be = e; bi = ia; }
                if (ex < xb) { ex = xb; ee = e; ei = ib; }
            }
        }
        first = 3;
    }
    for (int e0 = first; e0 < ne; e0 += 4) {
        // four edges' coordinates per memory round trip (see depth_row_mem)
        int2 x4[4];
#pragma unroll
        for (int k = 0; k < 4; k++) x4[k] = e0 + k < ne ? *(const int2 *)edge_piece(edges, np, ESZ / 4, e0 + k, 0, poly) : make_int2(0, 0);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int ia, xa, ib, xb;
            if (e0 + k < ne && edge_row(x4[k].x, x4[k].y, y, ia, xa, ib, xb)) {
                if (xa < bx) { bx = xa; be = e0 + k; bi = ia; }
                if (ex < xb) { ex = xb; ee = e0 + k; ei = ib; }
            }
        }
    }
    if (bx == INT_MAX) return false;
    int rb[ESZ], re[ESZ];
#pragma unroll
    for (int q = 0; q < ESZ / 4; q++) {
        const int4 vb = *edge_piece(edges, np, ESZ / 4, be, q, poly), ve = *edge_piece(edges, np, ESZ / 4, ee, q, poly);
        rb[4 * q] = vb.x; rb[4 * q + 1] = vb.y; rb[4 * q + 2] = vb.z; rb[4 * q + 3] = vb.w;
        re[4 * q] = ve.x; re[4 * q + 1] = ve.y; re[4 * q + 2] = ve.z; re[4 * q + 3] = ve.w;
    }
    S[0] = wadd(rb[2], wmul(bi, rb[3]));
    E[0] = wadd(re[2], wmul(ei, re[3]));
#pragma unroll
    for (int q = 1; q < NA; q++) {
        S[q] = wadd(rb[3 + q], wmul(bi, rb[3 + (NA - 1) + q]));
        E[q] = wadd(re[3 + q], wmul(ei, re[3 + (NA - 1) + q]));
    }
    beg = bx;
    end = ex;
    return true;
}

// The same row in two steps, for the draw-order fill: first which edges hold the row's two ends,
// with the depth there (everything the depth tests need, from the depth pieces of the edges:
// `p0` holds those of edges 0..2), then -- only for rows that get a colour -- the other
// interpolated values at the two ends.
struct RowOwners { int beg, end, be, bi, ee, ei; };

template <int NA>
__device__ __forceinline__ bool row_owners(const int *__restrict__ edges, int np, int poly, int ne, int y, const int4 (&p0)[3],
                                           RowOwners &o, int &zs, int &ze)
{
    constexpr int ESZ = edge_ints(NA);
    int bx = INT_MAX, ex = INT_MIN, be = 0, bi = 0, ee = 0, ei = 0;
    int bz0 = 0, bzi = 0, ez0 = 0, ezi = 0;
#pragma unroll
    for (int e = 0; e < 3; e++) {
        int ia, xa, ib, xb;
        if (e < ne && edge_row(p0[e].x, p0[e].y, y, ia, xa, ib, xb)) {
            if (xa < bx) { bx = xa; be = e; bi = ia; bz0 = p0[e].z; bzi = p0[e].w; }
            if (ex < xb) { ex = xb; ee = e; ei = ib; ez0 = p0[e].z; ezi = p0[e].w; }
        }
    }
    for (int e = 3; e < ne; e++) {
        const int4 r = *edge_piece(edges, np, ESZ / 4, e, 0, poly);
        int ia, xa, ib, xb;
        if (!edge_row(r.x, r.y, y, ia, xa, ib, xb)) continue;
        if (xa < bx) { bx = xa; be = e; bi = ia; bz0 = r.z; bzi = r.w; }
        if (ex < xb) { ex = xb; ee = e; ei = ib; ez0 = r.z; ezi = r.w; }
    }
    if (bx == INT_MAX) return false;
    zs = wadd(bz0, wmul(bi, bzi));
    ze = wadd(ez0, wmul(ei, ezi));
    o.beg = bx; o.end = ex; o.be = be; o.bi = bi; o.ee = ee; o.ei = ei;
    return true;
}

// S[1..NA-1], E[1..NA-1] of row_ends() for the owners found by row_owners()
template <int NA>
__device__ __forceinline__ void row_attrs(const int *__restrict__ edges, int np, int poly, const RowOwners &o, int (&S)[NA],
                                          int (&E)[NA])
{
    constexpr int ESZ = edge_ints(NA);
    int rb[ESZ], re[ESZ];
#pragma unroll
    for (int q = 1; q < ESZ / 4; q++) {
        const int4 vb = *edge_piece(edges, np, ESZ / 4, o.be, q, poly), ve = *edge_piece(edges, np, ESZ / 4, o.ee, q, poly);
        rb[4 * q] = vb.x; rb[4 * q + 1] = vb.y; rb[4 * q + 2] = vb.z; rb[4 * q + 3] = vb.w;
        re[4 * q] = ve.x; re[4 * q + 1] = ve.y; re[4 * q + 2] = ve.z; re[4 * q + 3] = ve.w;
    }
#pragma unroll
    for (int q = 1; q < NA; q++) {
        S[q] = wadd(rb[3 + q], wmul(o.bi, rb[3 + (NA - 1) + q]));
        E[q] = wadd(re[3 + q], wmul(o.ei, re[3 + (NA - 1) + q]));
    }
}

// (depth, draw index) packed for a signed 64-bit minimum: the depth is the signed high
// word, and between equal depths the low word compares as an unsigned magnitude.
__device__ __forceinline__ long long depth_key(int z, unsigned id)
{
    return (long long)(((unsigned long long)(unsigned)z << 32) | id);
}

// Depth span of one row from up to three edges held in registers (the depth pass of
// triangles), or from memory for polygons with more edges.  Returns false if no edge
// touches the row.  zs = depth accumulator at beg (16.16), zi = per-pixel increment
// (GraphicsPoly.cpp:352-353; the flat fill skips the divide when end == beg, which
// leaves the single pixel's depth unchanged: GraphicsPoly.cpp:273).
struct Edge4 { int xy1, xy2, z0, zi; };      // xy2: the packed deltas (edge_pack)

__device__ __forceinline__ bool depth_row3(const Edge4 (&ed)[3], int ne, int y, int &beg, int &end, int &zs, int &zi)
{
    int bx = INT_MAX, ex = INT_MIN;
    int bz0 = 0, bzi = 0, bi = 0, ez0 = 0, ezi = 0, ei = 0;       // the owning edges' depth walk and point index
#pragma unroll
    for (int e = 0; e < 3; e++) {
        int ia, xa, ib, xb;
        const bool hit = edge_row(ed[e].xy1, ed[e].xy2, y, ia, xa, ib, xb) && e < ne;
        const bool tb = hit && xa < bx, te = hit && ex < xb;
        bx = tb ? xa : bx; bz0 = tb ? ed[e].z0 : bz0; bzi = tb ? ed[e].zi : bzi; bi = tb ? ia : bi;
        ex = te ? xb : ex; ez0 = te ? ed[e].z0 : ez0; ezi = te ? ed[e].zi : ezi; ei = te ? ib : ei;
    }
    if (bx == INT_MAX) return false;
    const int sz = wadd(bz0, wmul(bi, bzi)), ez = wadd(ez0, wmul(ei, ezi));
    int span = ex - bx;
    if (span == 0) span = 1;
    beg = bx; end = ex; zs = sz;
    zi = tdivr(wsub(ez, sz), drcp(span));
    return true;
}

// The same from edges with their per-edge constants (edge_row_pre); an absent edge is one
// that no row hits.  The span's reciprocal comes from the table.
struct EdgeP { int xy1, P, z0, zi; double hr; };

__device__ __forceinline__ bool depth_row3_pre(const EdgeP (&ed)[3], int y, const double2 *lut, int &beg, int &end, int &zs,
                                               int &zi)
{
    int bx = INT_MAX, ex = INT_MIN;
    int bz0 = 0, bzi = 0, bi = 0, ez0 = 0, ezi = 0, ei = 0;
#pragma unroll
    for (int e = 0; e < 3; e++) {
        int ia, xa, ib, xb;
        const bool hit = edge_row_pre(ed[e].xy1, ed[e].P, ed[e].hr, y, ia, xa, ib, xb);
        const bool tb = hit && xa < bx, te = hit && ex < xb;
        bx = tb ? xa : bx; bz0 = tb ? ed[e].z0 : bz0; bzi = tb ? ed[e].zi : bzi; bi = tb ? ia : bi;
        ex = te ? xb : ex; ez0 = te ? ed[e].z0 : ez0; ezi = te ? ed[e].zi : ezi; ei = te ? ib : ei;
    }
    if (bx == INT_MAX) return false;
    const int sz = wadd(bz0, wmul(bi, bzi)), ez = wadd(ez0, wmul(ei, ezi));
    int span = ex - bx;
    if (span == 0) span = 1;
    beg = bx; end = ex; zs = sz;
    zi = tdivp(wsub(ez, sz), rcp_pair(lut, span));
    return true;
}

// The row's two ends alone (a polygon that lies wholly behind what its tile holds: its
// fragments are counted, nothing else of it is needed).
__device__ __forceinline__ bool extent_row3_pre(const EdgeP (&ed)[3], int y, int &beg, int &end)
{
    int bx = INT_MAX, ex = INT_MIN;
#pragma unroll
    for (int e = 0; e < 3; e++) {
        int ia, xa, ib, xb;
        const bool hit = edge_row_pre(ed[e].xy1, ed[e].P, ed[e].hr, y, ia, xa, ib, xb);
        bx = hit && xa < bx ? xa : bx;
        ex = hit && ex < xb ? xb : ex;
    }
    beg = bx; end = ex;
    return bx != INT_MAX;
}

__device__ __forceinline__ bool depth_row_mem(const int *__restrict__ edges, int npieces, int np, int poly, int ne, int y,
                                              int &beg, int &end, int &zs, int &zi)
{
    int bx = INT_MAX, ex = INT_MIN, sz = 0, ez = 0;
    // the coordinates of four edges at a time are fetched together, then the depth walk of the
    // two edges that hold the row's ends: polygons cut by a screen border have four to seven
    // edges, and one memory round trip per edge made the tiles along those borders the slowest
    int be = 0, bi = 0, ee = 0, ei = 0;
    for (int e0 = 0; e0 < ne; e0 += 4) {
        int2 x4[4];
#pragma unroll
        for (int k = 0; k < 4; k++) x4[k] = e0 + k < ne ? *(const int2 *)edge_piece(edges, np, npieces, e0 + k, 0, poly) : make_int2(0, 0);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int ia, xa, ib, xb;
            if (e0 + k < ne && edge_row(x4[k].x, x4[k].y, y, ia, xa, ib, xb)) {
                if (xa < bx) { bx = xa; be = e0 + k; bi = ia; }
                if (ex < xb) { ex = xb; ee = e0 + k; ei = ib; }
            }
        }
    }
    if (bx != INT_MAX) {
        const int2 zb = ((const int2 *)edge_piece(edges, np, npieces, be, 0, poly))[1], ze = ((const int2 *)edge_piece(edges, np, npieces, ee, 0, poly))[1];
        sz = wadd(zb.x, wmul(bi, zb.y));
        ez = wadd(ze.x, wmul(ei, ze.y));
    }
    if (bx == INT_MAX) return false;
    int span = ex - bx;
    if (span == 0) span = 1;
    beg = bx; end = ex; zs = sz;
    zi = tdivr(wsub(ez, sz), drcp(span));
    return true;
}

// One linear run of the perspective mapper (GraphicsPoly.cpp:614-640): exact texture coordinates
// at the run's two ends xs, xe (pixels from the screen centre) on screen row gy, and the
// per-pixel increments over `rspan` pixels (0: the run is a single pixel).
__device__ __forceinline__ void prp_run(const RasterArgs &a, int poly, int gy, int xs, int xe, int rspan, int &t0u, int &t0v,
                                        int &iu, int &iv)
{
    const int4 c0 = ((const int4 *)(a.prp + (size_t)poly * PRP_INTS))[0];
    const int4 c1 = ((const int4 *)(a.prp + (size_t)poly * PRP_INTS))[1];
    const int4 c2 = ((const int4 *)(a.prp + (size_t)poly * PRP_INTS))[2];
    const int lf = a.log_focus, lsc = c2.w;
    const int yp = gy - a.H / 2;
    const int xx = wadd(wmul(yp, c0.y) >> lf, c0.z), yy = wadd(wmul(yp, c1.x) >> lf, c1.y);
    const int zz = wadd(wmul(yp, c1.w) >> lf, c2.x);
    int t1u = 0, t1v = 0;
    t0u = 0; t0v = 0;
    {
        const int zzz = wadd(zz, wmul(xs, c1.z) >> lf);
        if (zzz != 0) {
            t0u = wadd(wshl(tdiv_any(wshl(wadd(xx, wmul(xs, c0.x) >> lf), lsc), zzz), 16), c2.y);
            t0v = wadd(wshl(tdiv_any(wshl(wadd(yy, wmul(xs, c0.w) >> lf), lsc), zzz), 16), c2.z);
        }
    }
    {
        const int zzz = wadd(zz, wmul(xe, c1.z) >> lf);
        if (zzz != 0) {
            t1u = wadd(wshl(tdiv_any(wshl(wadd(xx, wmul(xe, c0.x) >> lf), lsc), zzz), 16), c2.y);
            t1v = wadd(wshl(tdiv_any(wshl(wadd(yy, wmul(xe, c0.w) >> lf), lsc), zzz), 16), c2.z);
        }
    }
    iu = 0; iv = 0;
    if (rspan != 0) { const double rr = drcp(rspan); iu = tdivr(wsub(t1u, t0u), rr); iv = tdivr(wsub(t1v, t0v), rr); }
}

// Address of the texel that the textured fills read at pixel (gx, gy) of a polygon row
// [beg, end] with start values S and per-pixel increments I (GraphicsPoly.cpp:465-486 linear
// texture, :592-667 perspective texture); null for the fills that use the polygon's colour.
template <int NA>
__device__ __forceinline__ const int *texel_address(const RasterArgs &a, const int4 h, int poly, int gx, int gy, int beg,
                                                    int end, const int (&S)[NA], const int (&I)[NA])
{
    const int kind = (h.w >> 8) & 0xff, ptex = (h.w >> 16) & 0xff;
    const int kx = gx - beg;
    if (NA == 6 && kind == KIND_TEX_PRP) {
        // exact texture coordinates at the ends of 32-pixel runs, linear in between
        const int begp = beg - a.W / 2, endp = end - a.W / 2, xp = gx - a.W / 2;
        const int d = xp - begp;
        int xs, q;                                   // run start boundary, pixel offset in the run
        if (d <= 32) { xs = begp; q = d; }
        else { const int k32 = ((d - 1) >> 5) << 5; xs = begp + k32; q = xp - xs - 1; }
        int xe = xs + 32;
        if (xe > endp) xe = endp;
        const int rspan = d <= 32 ? xe - xs : xe - xs - 1;
        int t0u, t0v, iu, iv;
        prp_run(a, poly, gy, xs, xe, rspan, t0u, t0v, iu, iv);
        const int lg2 = a.tex_log[ptex];
        const int mask = wsub(wshl(1, lg2 + 16), 1);
        const int tu = wadd(t0u, wmul(q, iu)) & mask, tv = wadd(t0v, wmul(q, iv)) & mask;
        return a.tex_ptr[ptex] + (((tv >> 16) << lg2) + (tu >> 16));
    }
    if (NA == 6 && kind == KIND_TEX_LIN) {
        // GraphicsPoly.cpp:443-446,465,486: wrap, then fetch the texel
        const int lg2 = a.tex_log[ptex];
        const int mask = wsub(wshl(1, lg2 + 16), 1);
        const int tu = wadd(S[4 % NA], wmul(kx, I[4 % NA])) & mask;
        const int tv = wadd(S[5 % NA], wmul(kx, I[5 % NA])) & mask;
        return a.tex_ptr[ptex] + (((tv >> 16) << lg2) + (tu >> 16));
    }
    return nullptr;
}

// Colour of the pixel (gx, gy) of a polygon row [beg, end] with start values S and
// per-pixel increments I (GraphicsPoly.cpp:372 shaded, :465-491 linear texture,
// :592-675 perspective texture).  `kl` is the number of increments the light
// accumulators have taken when this pixel is drawn: gx - beg for the shaded fill, the
// number of earlier pixels of the span that passed the depth test for the textured
// fills (GraphicsPoly.cpp:485-493, 666-674).
template <int NA>
__device__ __forceinline__ int pixel_colour(const RasterArgs &a, const int4 h, int poly, int gx, int gy, int beg,
                                            int end, const int (&S)[NA], const int (&I)[NA], int kl)
{
    const int lr = wadd(S[1 % NA], wmul(kl, I[1 % NA])) >> 16;
    const int lg = wadd(S[2 % NA], wmul(kl, I[2 % NA])) >> 16;
    const int lb = wadd(S[3 % NA], wmul(kl, I[3 % NA])) >> 16;
    const int *t = texel_address<NA>(a, h, poly, gx, gy, beg, end, S, I);
    return cl_light(t ? *t : h.z, lr, lg, lb);
}

// Draw-order fill for textured batches whose light varies along spans.  There the
// reference advances the light accumulators only where the depth test passes
// (GraphicsPoly.cpp:485-493, 666-674), so a pixel's colour depends on what every earlier
// polygon left in the depth buffer along the same row.  Rows do not interact, so a
// workgroup owns a band of full-width rows (depth and colour in LDS), each wave owns
// every fourth row of the band, and each wave walks ALL polygons in draw order: it reads
// headers 64 at a time, keeps those that touch the band (ballot), and sweeps each of
// their rows 64 pixels per step -- depth test, ballot, prefix count of the passes, colour.
// No lists and no atomics: the order is the array order.  Runs only if the setup kernel
// raised *order_flag; the tile kernel then returns at once.
template <int NA>
__global__ void __launch_bounds__(256) ordered_kernel(RasterArgs a, int R)
{
    extern __shared__ int band_lds[];
    if (!(a.force_order || *a.order_flag == a.order_gen)) return;
    const int W = a.W;
    int *zt = band_lds, *ct = band_lds + R * W;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int y0 = max(blockIdx.x * R, a.band_y0), y1 = min(min(blockIdx.x * R + R, a.H), a.band_y1);
    if (y0 >= y1) return;
    const int npx = (y1 - y0) * W;
    for (int p = t; p < npx; p += 256) {
        const size_t g = (size_t)y0 * W + p;
        zt[p] = a.clear_pending ? INT_MAX : a.zbuf[g];
        ct[p] = a.clear_pending ? a.clear_colour : a.cbuf[g];
    }
    __syncthreads();

    unsigned long long frags = 0;
    for (int base = 0; base < a.np; base += 64) {
        const int mypoly = base + lane;
        int4 h = make_int4(0, 0, 0, 0);
        if (mypoly < a.np) {
            const int2 rc = ((const int2 *)a.hdr)[mypoly];
            // the whole header (colour, edge count, kind, texture) is at the front of the polygon's record
            if (rc.x != 0 && (rc.x & 0xffff) < y1 && ((rc.x >> 16) & 0xffff) > y0) h = *record_header(a.edges, edge_ints(NA) / 4, mypoly);
        }
        const bool ov = (h.w & 0xff) != 0 && (h.x & 0xffff) < y1 && ((h.x >> 16) & 0xffff) > y0;
        unsigned long long mask = __ballot(ov);
        while (mask) {
            const int src = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const int4 bh = make_int4(__shfl(h.x, src), __shfl(h.y, src), __shfl(h.z, src), __shfl(h.w, src));
            const int poly = base + src;
            const int ne = bh.w & 0xff, kind = (bh.w >> 8) & 0xff;
            const int ra = max(bh.x & 0xffff, y0), rb = min((bh.x >> 16) & 0xffff, y1);
            // first row of this wave (rows y0+wv, y0+wv+4, ...) at or after ra
            int y = y0 + wv + (((ra - y0 - wv + 3) >> 2) << 2);
            if (ra - y0 < wv) y = y0 + wv;
            for (; y < rb; y += 4) {
                int S[NA], E[NA], I[NA], beg = 0, end = 0;
                if (!row_ends<NA>(a.edges, a.np, poly, ne, y, beg, end, S, E)) continue;
                int span = end - beg;
                if (span == 0) span = 1;
                const double rs = drcp(span);
#pragma unroll
                for (int q = 0; q < NA; q++) I[q] = tdivr(wsub(E[q], S[q]), rs);
                if (lane == 0) frags += (unsigned long long)(end - beg + 1);
                int *zrow = zt + (y - y0) * W, *crow = ct + (y - y0) * W;
                int carry = 0;
                for (int xb = beg; xb <= end; xb += 64) {
                    const int x = xb + lane;
                    const bool in = x <= end;
                    const int k = x - beg;
                    const int zi = wadd(S[0], wmul(k, I[0])) >> 16;
                    const bool pass = in && zrow[x < W ? x : W - 1] > zi;
                    const unsigned long long pm = __ballot(pass);
                    if (pass) {
                        const int nb = carry + __popcll(pm & ((1ull << lane) - 1ull));
                        int c = bh.z;
                        if (NA > 1 && kind != KIND_FLAT)
                            c = pixel_colour<NA>(a, bh, poly, x, y, beg, end, S, I, kind >= KIND_TEX_LIN ? nb : k);
                        zrow[x] = zi;
                        crow[x] = c;
                    }
                    carry += __popcll(pm);
                }
            }
        }
    }
    __syncthreads();
    for (int p = t; p < npx; p += 256) {
        const size_t g = (size_t)y0 * W + p;
        a.zbuf[g] = zt[p];
        a.cbuf[g] = ct[p];
    }
    if (lane == 0 && frags) atomicAdd(&(a.status + (blockIdx.x % STATUS_SLOTS))->fragments, frags);
}

// ---- draw-order fill from per-band polygon lists
//
// The kernel above reads every header in every band.  For batches of millions of polygons the
// bands get lists instead: for every band of R rows, the slots of the polygons that have rows
// in it, in slot (= draw) order.  They come from a stable counting sort over (slot, band)
// pairs.  A workgroup takes a chunk of BAND_CHUNK consecutive slots and sets, in an LDS bit
// matrix [band][slot in chunk], the bit of every (polygon, band) pair; a row of that matrix
// read from bit 0 upwards is the chunk's part of the band's list, already in order.
//   band_bin_kernel<false>   bit matrix -> cnt[chunk][band] = bits set in the row
//   band_scan_kernel         cnt -> exclusive prefix along the chunks; every band gets its
//                            list's place (base, total) from one running allocation
//   band_bin_kernel<true>    bit matrix again -> list[base[band] + cnt[chunk][band] + rank]
// All of them return at once unless the batch is to be filled in draw order.

constexpr int BAND_CHUNK = 1024;
constexpr int BAND_THREADS = 1024;          // the bit matrix is large (two workgroups a CU): many waves per workgroup
constexpr int BAND_WORDS = BAND_CHUNK / 32;
constexpr int BAND_PITCH = BAND_WORDS + 1;       // LDS words between two bands' rows (bank spread)

template <bool SCATTER>
__global__ void __launch_bounds__(BAND_THREADS) band_bin_kernel(BandArgs a, int nchunks)
{
    extern __shared__ unsigned band_bits[];
    if (!(a.force_order || *a.order_flag == a.order_gen)) return;
    const int t = threadIdx.x;
    // The grid is capped (the bit matrix is large: few workgroups fit a CU) and strides over the
    // chunks, so a batch that is not filled in draw order costs a few hundred empty workgroups.
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        for (int i = t; i < a.nb * BAND_PITCH; i += BAND_THREADS) band_bits[i] = 0u;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BAND_CHUNK / BAND_THREADS; k++) {
            const int idx = k * BAND_THREADS + t, slot = chunk * BAND_CHUNK + idx;
            if (slot < a.np) {
                const int2 h = ((const int2 *)a.hdr)[slot];
                if (h.x != 0) {
                    const int miny = h.x & 0xffff, maxy = (h.x >> 16) & 0xffff;
                    const int b0 = miny / a.R, b1 = min((maxy - 1) / a.R, a.nb - 1);
                    for (int b = b0; b <= b1; b++) atomicOr(&band_bits[b * BAND_PITCH + (idx >> 5)], 1u << (idx & 31));
                }
            }
        }
        __syncthreads();
        for (int b = t; b < a.nb; b += BAND_THREADS) {
            const unsigned *row = band_bits + b * BAND_PITCH;
            if (!SCATTER) {
                unsigned c = 0;
#pragma unroll 8
                for (int k = 0; k < BAND_WORDS; k++) c += __popc(row[k]);
                a.cnt[(size_t)chunk * a.nb + b] = c;
            } else {
                unsigned long long pos = (unsigned long long)a.base[b] + a.cnt[(size_t)chunk * a.nb + b];
                for (int k = 0; k < BAND_WORDS; k++) {
                    unsigned w = row[k];
                    while (w) {
                        const int bit = __ffs((int)w) - 1;
                        w &= w - 1;
                        if (pos < a.capacity) a.list[pos] = (unsigned)(chunk * BAND_CHUNK + k * 32 + bit);
                        pos++;
                    }
                }
            }
        }
        __syncthreads();
    }
}

// One workgroup per 64 bands (a lane per band), its 16 waves each a stretch of the chunks.
__global__ void __launch_bounds__(1024) band_scan_kernel(BandArgs a, int nchunks)
{
    __shared__ unsigned seg[16][64];
    if (!(a.force_order || *a.order_flag == a.order_gen)) return;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int b = blockIdx.x * 64 + lane;
    const bool on = b < a.nb;
    const int per = (nchunks + 15) / 16, c0 = w * per, c1 = min(c0 + per, nchunks);
    // (eight loads in flight: a lane's chain of dependent loads is what this kernel lasts)
    unsigned s = 0;
    if (on) {
        int c = c0;
        for (; c + 8 <= c1; c += 8) {
            unsigned v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = a.cnt[(size_t)(c + q) * a.nb + b];
#pragma unroll
            for (int q = 0; q < 8; q++) s += v[q];
        }
        for (; c < c1; c++) s += a.cnt[(size_t)c * a.nb + b];
    }
    seg[w][lane] = s;
    __syncthreads();
    unsigned pre = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const unsigned v = seg[k][lane];
        pre += k < w ? v : 0u;
        tot += v;
    }
    if (w == 0) {
        unsigned incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        const unsigned all = __shfl(incl, 63);
        unsigned g = 0;
        if (lane == 0 && all) {
            g = atomicAdd(a.alloc, all);
            const unsigned long long end = (unsigned long long)g + all;
            if (end > a.capacity)
                atomicAdd(&a.status->list_overflow, end - (a.capacity > g ? a.capacity : (unsigned long long)g));
        }
        g = __shfl(g, 0);
        if (on) { a.base[b] = g + incl - tot; a.total[b] = tot; }
    }
    __syncthreads();
    if (on) {
        unsigned run = pre;
        int c = c0;
        for (; c + 8 <= c1; c += 8) {
            unsigned v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = a.cnt[(size_t)(c + q) * a.nb + b];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                a.cnt[(size_t)(c + q) * a.nb + b] = run;
                run += v[q];
            }
        }
        for (; c < c1; c++) {
            const size_t i = (size_t)c * a.nb + b;
            const unsigned v = a.cnt[i];
            a.cnt[i] = run;
            run += v;
        }
    }
}

// The draw-order fill of one row from its band's list: one workgroup per row (depth and colour
// of the row in LDS), its NW waves taking groups of 64 list entries in turn (group g goes to
// wave g mod NW).  For a group:
//  - all 64 lanes find their polygon's span on the row: its two ends and the depth there, from
//    the 64-byte run of the polygon record that was fetched a group ahead.  This does not
//    depend on what is in the row, so the waves do it side by side;
//  - still outside any turn, every span is tested against the depth values the row holds at
//    that moment (a long span against a coarse table: the largest depth of each aligned
//    64-pixel chunk).  Depth values only get smaller, so a span that passes nowhere now passes
//    nowhere in its turn, and such a span changes nothing: it drops out.  The others fetch what
//    their colours need (light and texture coordinates at the row's ends) while they wait;
//  - then the group waits for its turn at the row's depth values (LDS word turn[0]).  Spans
//    of up to 16 pixels are depth-tested one per lane in rounds: a lane may go once no earlier
//    polygon of the 64 that is still waiting has a span near its own.  Lanes claim the 8- or
//    16-pixel cells that their span touches with their lane number (an LDS minimum); a lane
//    that holds all of its cells has no earlier neighbour left, tests its pixels, keeps a bit
//    mask of the passes and frees its cells.  The lowest waiting lane always holds its cells:
//    every round makes progress.  A longer span is swept between the rounds of the short spans
//    before it and after it, chunk by chunk (chunks the coarse table rejects are skipped), a
//    pixel per lane; the pass mask of a chunk with passes is recorded;
//  - with the depth values updated the next group may start on them; this one computes the
//    colours of its passes (texel loads, light from the pass count: GraphicsPoly.cpp:485-493)
//    and writes them when its turn at the row's colours has come (turn[1]), in the order the
//    depth tests were made;
//  - a group with more long-span chunks or rounds than it can record takes the colour turn
//    early, writes what is pending and draws the rest with both turns held.
__device__ __forceinline__ void wait_turn(const unsigned *word, unsigned k)
{
    while (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != k) __builtin_amdgcn_s_sleep(2);
}
__device__ __forceinline__ void pass_turn(unsigned *word, unsigned k, int lane)
{
    LDS_WAVE_SYNC();
    if (lane == 0) __hip_atomic_store(word, k + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int NA, int NW>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(3, 3))) ordered_list_kernel(RasterArgs a, BandArgs ba, int cshift)
{
    constexpr int ESZ = edge_ints(NA);
    constexpr int SHORT = 16;                 // spans of up to SHORT pixels are drawn one per lane
    constexpr int LONG_PRE = 4;               // chunks of long spans whose colours are worked out before the colour turn
    extern __shared__ int row_lds[];
    if (!(a.force_order || *a.order_flag == a.order_gen)) return;
    const int W = a.W, R = ba.R;
    const int ncell = (W >> cshift) + 1;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    // Workgroups go to the eight XCDs in turn, and each XCD has its own L2: the R workgroups that
    // draw the rows of one band read the same polygon records, so they take indices that are
    // eight apart (same XCD, started together) and share those records through one L2.
    const int grp = blockIdx.x / (8 * R), in = blockIdx.x % (8 * R);
    const int band = grp * 8 + in % 8, y = band * R + in / 8;
    if (blockIdx.x == 0 && t == 0) *ba.alloc = 0u;          // every list has its place by now: ready for the next batch
    if (band >= ba.nb) return;
    if (y >= a.H || y < a.band_y0 || y >= a.band_y1) return;
    int *zrow = row_lds, *crow = row_lds + W;
    unsigned *cells = (unsigned *)(row_lds + 2 * W), *turn = cells + ncell;
    int *zcoarse = (int *)(turn + 2);           // largest depth of each aligned 64-pixel chunk of the row (never too small)
    const int nchunk = (W + 63) >> 6;
    const size_t g0 = (size_t)y * W;
    for (int p = t; p < W; p += 64 * NW) {
        zrow[p] = a.clear_pending ? INT_MAX : a.zbuf[g0 + p];
        crow[p] = a.clear_pending ? a.clear_colour : a.cbuf[g0 + p];
    }
    for (int p = t; p < ncell + 2; p += 64 * NW) cells[p] = p < ncell ? ~0u : 0u;       // the two turn words start at 0
    __syncthreads();
    for (int ch = wv; ch < nchunk; ch += NW) {
        const int x = (ch << 6) + lane;
        int m = x < W ? zrow[x] : INT_MIN;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) m = max(m, __shfl_xor(m, d));
        if (lane == 0) zcoarse[ch] = m;
    }
    __syncthreads();

    const unsigned long long lb = ba.base[band];
    unsigned ln = ba.total[band];
    if (lb + ln > ba.capacity) ln = ba.capacity > lb ? (unsigned)(ba.capacity - lb) : 0u;     // overflow is reported by the scan
    const unsigned ngroups = (ln + 63u) / 64u;
    unsigned long long frags = 0;
    const int4 zero4 = make_int4(0, 0, 0, 0);
    // the list entry, header and edge coordinates of a wave's next group are fetched while it works on this one
    auto fetch = [&](unsigned k, int &poly, int4 &h, int4 (&p0)[3]) __attribute__((always_inline)) {
        const unsigned ei = k * 64u + (unsigned)lane;
        poly = (k < ngroups && ei < ln) ? (int)ba.list[lb + ei] : -1;
        if ((unsigned)poly >= (unsigned)a.np) poly = -1;
        h = zero4;
#pragma unroll
        for (int e = 0; e < 3; e++) p0[e] = zero4;
        if (poly >= 0) {
            // header and the depth pieces of edges 0..2: one 64-byte run of the polygon record
            h = *record_header(a.edges, ESZ / 4, poly);
#pragma unroll
            for (int e = 0; e < 3; e++) p0[e] = *edge_piece(a.edges, a.np, ESZ / 4, e, 0, poly);
        }
    };
    int poly_n;
    int4 h_n;
    int4 p0_n[3];
    fetch((unsigned)wv, poly_n, h_n, p0_n);
    for (unsigned k = (unsigned)wv; k < ngroups; k += NW) {
        // ---- the spans of this group's polygons on the row: ends and depth only.  The other
        // interpolated values are fetched after the depth tests, for the spans that get a colour.
        const int poly = poly_n;
        const int4 h = h_n;
        const int4 p0[3] = {p0_n[0], p0_n[1], p0_n[2]};
        fetch(k + NW, poly_n, h_n, p0_n);
        const int ne = h.w & 0xff, kind = (h.w >> 8) & 0xff;
        int S[NA], E[NA], I[NA], beg = 0, end = -1;
#pragma unroll
        for (int q = 0; q < NA; q++) { S[q] = 0; E[q] = 0; I[q] = 0; }
        RowOwners ow{};
        bool hit = poly >= 0 && ne != 0 && y >= (h.x & 0xffff) && y < ((h.x >> 16) & 0xffff);
        if (hit) hit = row_owners<NA>(a.edges, a.np, poly, ne, y, p0, ow, S[0], E[0]);
        if (hit) {
            beg = ow.beg; end = ow.end;
            int span = end - beg;
            if (span == 0) span = 1;
            I[0] = tdivr(wsub(E[0], S[0]), drcp(span));
            frags += (unsigned long long)(end - beg + 1);
        }
        int4 hc = h;
        int ckind = kind;
        bool have_attr = false;
        // the values a span's colours need: light and texture coordinates at its ends, increments.
        // A span of up to 32 pixels is one linear run of the perspective mapper (GraphicsPoly.cpp:626):
        // its texture coordinates are start + pixel * increment, the linear mapper's form, so
        // the run is worked out once here and the colours below know only that form.
        auto attrs = [&](bool want) __attribute__((always_inline)) {
            want = want && hit && !have_attr;
            if (want) {
                row_attrs<NA>(a.edges, a.np, poly, ow, S, E);
                int span = end - beg;
                if (span == 0) span = 1;
                const double rs = drcp(span);
#pragma unroll
                for (int q = 1; q < NA; q++) I[q] = tdivr(wsub(E[q], S[q]), rs);
                have_attr = true;
            }
            if (NA == 6 && want && kind == KIND_TEX_PRP && end - beg < SHORT) {
                prp_run(a, poly, y, beg - W / 2, end - W / 2, end - beg, S[4 % NA], S[5 % NA], I[4 % NA], I[5 % NA]);
                hc.w = (h.w & ~0xff00) | (KIND_TEX_LIN << 8);
                ckind = KIND_TEX_LIN;
            }
        };
        // ---- before the turn: depth values only get smaller, so a span none of whose pixels
        // passes against what the row holds now (earlier groups may still be lowering it) cannot
        // pass in its turn either, and a span without a pass changes nothing: it drops out here.
        // The others fetch what their colours need while they wait for the turn.
        if (hit && end - beg < SHORT) {
            const int n = end - beg;
            int z = S[0];
            bool may = false;
#pragma unroll
            for (int b4 = 0; b4 < SHORT / 4; b4++) {
                if (n >= 4 * b4) {
                    int zo[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) zo[q] = zrow[beg + min(4 * b4 + q, n)];
#pragma unroll
                    for (int q = 0; q < 4; q++) may = may || zo[q] > (wadd(z, wmul(q, I[0])) >> 16);
                }
                z = wadd(z, wshl(I[0], 2));
            }
            hit = may;
        }
        for (unsigned long long lm = __ballot(hit && end - beg >= SHORT); lm; lm &= lm - 1ull) {
            const int src = __ffsll((long long)lm) - 1;
            const int bbeg = __builtin_amdgcn_readlane(beg, src), bend = __builtin_amdgcn_readlane(end, src);
            const int bS0 = __builtin_amdgcn_readlane(S[0], src), bI0 = __builtin_amdgcn_readlane(I[0], src);
            const long long zlast = (long long)bS0 + (long long)(bend - bbeg) * bI0;
            bool any = zlast != (long long)(int)zlast;          // a wrap along the span: no coarse test
            const int ch0 = bbeg >> 6, ch1 = bend >> 6;
            for (int cb = ch0; cb <= ch1 && !any; cb += 64) {
                const int ch = cb + lane;
                bool need = false;
                if (ch <= ch1) {
                    const int xa = max(bbeg, ch << 6), xe = min(bend, (ch << 6) + 63);
                    const int za = wadd(bS0, wmul(xa - bbeg, bI0)) >> 16, ze = wadd(bS0, wmul(xe - bbeg, bI0)) >> 16;
                    need = min(za, ze) < zcoarse[ch];
                }
                any = __ballot(need) != 0ull;
            }
            if (!any && lane == src) hit = false;
        }
        const unsigned long long hits = __ballot(hit);
        const unsigned long long longs = __ballot(hit && end - beg >= SHORT);
        attrs(true);

        const int c0 = beg >> cshift, c1 = end >> cshift;          // a short span touches at most three cells
        // One round of the short spans in `rem`: lanes claim their cells; those that hold them all
        // test their pixels against the row's depth values (GraphicsPoly.cpp:469-472; the values
        // are read together, one LDS round trip, then compared and lowered) and free the cells.
        // Returns the lanes that went; `pass` gets a bit per pixel that passed.
        auto depth_round = [&](unsigned long long rem, unsigned &pass) __attribute__((always_inline)) -> unsigned long long {
            const bool mine = (rem >> lane) & 1ull;
            if (mine) {
                atomicMin(&cells[c0], (unsigned)lane);
                if (c1 > c0) atomicMin(&cells[c0 + 1], (unsigned)lane);
                if (c1 > c0 + 1) atomicMin(&cells[c1], (unsigned)lane);
            }
            LDS_WAVE_SYNC();
            bool go = false;
            if (mine)
                go = cells[c0] == (unsigned)lane && (c1 <= c0 || cells[c0 + 1] == (unsigned)lane) &&
                     (c1 <= c0 + 1 || cells[c1] == (unsigned)lane);
            LDS_WAVE_SYNC();
            if (go) {
                // four pixels at a time; only the lanes whose span reaches a block touch the row
                const int n = end - beg;
                int z = S[0];
#pragma unroll
                for (int b4 = 0; b4 < SHORT / 4; b4++) {
                    if (n >= 4 * b4) {
                        int zo[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) zo[q] = zrow[beg + min(4 * b4 + q, n)];
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int zi = wadd(z, wmul(q, I[0])) >> 16;
                            if (4 * b4 + q <= n && zo[q] > zi) { zrow[beg + 4 * b4 + q] = zi; pass |= 1u << (4 * b4 + q); }
                        }
                    }
                    z = wadd(z, wshl(I[0], 2));
                }
                cells[c0] = ~0u;
                if (c1 > c0) cells[c0 + 1] = ~0u;
                if (c1 > c0 + 1) cells[c1] = ~0u;
            }
            LDS_WAVE_SYNC();
            return __ballot(go);
        };
        // Colours of the pixels in `pass`: light has advanced once per earlier pass of the span
        // (textured fills, GraphicsPoly.cpp:485-493), once per pixel (shaded).  Four pixels at a
        // time, so that their texel loads are in flight together.
        int col[SHORT];
        auto colours = [&](unsigned pass) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < SHORT; q++) col[q] = hc.z;
            // every texel load of the span first (blocks of four pixels that no lane needs are skipped),
            // so that they are in flight together; then the light
#pragma unroll
            for (int g = 0; g < SHORT / 4; g++) {
                if (__ballot((pass >> (4 * g)) & 0xfu) == 0ull) continue;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (((pass >> (4 * g + q)) & 1u) && ckind != KIND_FLAT) {
                        const int *tp = texel_address<NA>(a, hc, poly, beg + 4 * g + q, y, beg, end, S, I);
                        if (tp) col[4 * g + q] = *tp;
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < SHORT / 4; g++) {
                if (__ballot((pass >> (4 * g)) & 0xfu) == 0ull) continue;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int px = 4 * g + q;
                    if (NA > 1 && ckind != KIND_FLAT) {
                        const int kl = ckind >= KIND_TEX_LIN ? __popc(pass & ((1u << px) - 1u)) : px;
                        col[px] = cl_light(col[px], wadd(S[1 % NA], wmul(kl, I[1 % NA])) >> 16,
                                           wadd(S[2 % NA], wmul(kl, I[2 % NA])) >> 16,
                                           wadd(S[3 % NA], wmul(kl, I[3 % NA])) >> 16);
                    }
                }
            }
        };
        auto write_colours = [&](unsigned pass) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < SHORT; q++)
                if ((pass >> q) & 1u) crow[beg + q] = col[q];
        };

        // ---- depth phase, in draw order, under the row's depth turn.  Short spans go in rounds
        // (sequence number `seq` per round); a long span is swept in 64-pixel chunks aligned to the
        // row: a coarse table (largest depth of each chunk) rejects whole chunks, the others are
        // tested a pixel per lane.  Nothing is coloured here: a short span keeps its pass mask, a
        // chunk of a long span with passes is recorded in the registers of lane `nent` (chunk
        // start, source lane and sequence number, passes before the chunk, pass mask).
        int ent_xb = 0, ent_tag = 0, ent_carry = 0;
        unsigned ent_lo = 0u, ent_hi = 0u;
        int nent = 0, seq = 0, my_seq = -1;
        unsigned pass = 0u;
        unsigned long long long_seqs = 0ull;          // bit s: sequence number s is a long span
        unsigned long long long_used = 0ull;          // lanes whose long span has a recorded chunk
        bool immediate = false;                       // colours written at once, both turns held
        const unsigned long long lt_mask = (1ull << lane) - 1ull;
        auto RL = [&](int v, int l) __attribute__((always_inline)) { return __builtin_amdgcn_readlane(v, l); };
        // a long span's data, read from the lane that holds it (wave-uniform values), and the
        // colour of its pixel x after `nbk` earlier passes
        struct Span { int4 h; int poly, beg, end; int S[NA], I[NA]; };
        auto span_of = [&](int src) __attribute__((always_inline)) -> Span {
            Span sp;
            sp.h = make_int4(RL(h.x, src), RL(h.y, src), RL(h.z, src), RL(h.w, src));
            sp.poly = RL(poly, src); sp.beg = RL(beg, src); sp.end = RL(end, src);
#pragma unroll
            for (int q = 0; q < NA; q++) { sp.S[q] = RL(S[q], src); sp.I[q] = RL(I[q], src); }
            return sp;
        };
        auto span_colour = [&](const Span &sp, int x, int nbk) __attribute__((always_inline)) -> int {
            const int bkind = (sp.h.w >> 8) & 0xff;
            if (NA == 1 || bkind == KIND_FLAT) return sp.h.z;
            return pixel_colour<NA>(a, sp.h, sp.poly, x, y, sp.beg, sp.end, sp.S, sp.I, bkind >= KIND_TEX_LIN ? nbk : x - sp.beg);
        };
        int lcol[LONG_PRE];
        // colour writes of everything recorded so far, in sequence order (under the colour turn)
        auto flush = [&](bool pre) __attribute__((always_inline)) {
            int e = 0;
            for (int sq = 0; sq < seq; sq++) {
                if ((long_seqs >> sq) & 1ull) {
                    while (e < nent) {
                        const int tag = RL(ent_tag, e);
                        if ((tag >> 8) != sq) break;
                        const unsigned long long pm = (unsigned long long)(unsigned)RL((int)ent_lo, e) |
                                                      ((unsigned long long)(unsigned)RL((int)ent_hi, e) << 32);
                        const int xb = RL(ent_xb, e), cr = RL(ent_carry, e);
                        if (pre && e < LONG_PRE) {
                            int c = 0;
#pragma unroll
                            for (int q = 0; q < LONG_PRE; q++) c = e == q ? lcol[q] : c;
                            if ((pm >> lane) & 1ull) crow[xb + lane] = c;
                        } else {
                            const Span sp = span_of(tag & 0xff);
                            if ((pm >> lane) & 1ull) crow[xb + lane] = span_colour(sp, xb + lane, cr + __popcll(pm & lt_mask));
                        }
                        e++;
                    }
                } else if (my_seq == sq) {
                    write_colours(pass);
                }
                LDS_WAVE_SYNC();
            }
        };
        // more long spans or rounds than the records hold: take the colour turn as well, write what
        // is pending, and draw the rest of the group with its colours at once
        auto go_immediate = [&]() __attribute__((always_inline)) {
            wait_turn(&turn[1], k);
            attrs(true);
            colours(pass);
            flush(false);
            immediate = true;
            pass = 0u; my_seq = -1; nent = 0; seq = 0; long_seqs = 0ull;
        };

        wait_turn(&turn[0], k);
        unsigned long long todo = hits;
        while (todo) {
            const unsigned long long ltd = longs & todo;
            const int first_long = ltd ? __ffsll((long long)ltd) - 1 : 64;
            const unsigned long long sub = first_long == 64 ? todo : todo & ((1ull << first_long) - 1ull);
            for (unsigned long long rem = sub; rem;) {
                if (!immediate && seq >= 63) go_immediate();
                unsigned long long went;
                if (!immediate) {
                    went = depth_round(rem, pass);
                    if ((went >> lane) & 1ull) my_seq = seq;
                    seq++;
                } else {
                    unsigned p1 = 0u;
                    went = depth_round(rem, p1);
                    colours(p1);
                    write_colours(p1);
                    LDS_WAVE_SYNC();
                }
                rem &= ~went;
            }
            todo &= ~sub;
            if (first_long < 64) {
                const int src = first_long;
                todo &= ~(1ull << src);
                if (!immediate && seq >= 63) go_immediate();
                Span sp = span_of(src);
                const int bbeg = sp.beg, bend = sp.end, bS0 = sp.S[0], bI0 = sp.I[0];
                // without a 32-bit wrap along the span the depth runs monotonically: a chunk's
                // smallest depth is at one of its ends
                const long long zlast = (long long)bS0 + (long long)(bend - bbeg) * bI0;
                const bool mono = zlast == (long long)(int)zlast;
                const int ch0 = bbeg >> 6, ch1 = bend >> 6;
                int carry = 0;
                for (int cb = ch0; cb <= ch1; cb += 64) {
                    const int ch = cb + lane;
                    bool need = ch <= ch1;
                    if (need && mono) {
                        const int xa = max(bbeg, ch << 6), xe = min(bend, (ch << 6) + 63);
                        const int za = wadd(bS0, wmul(xa - bbeg, bI0)) >> 16, ze = wadd(bS0, wmul(xe - bbeg, bI0)) >> 16;
                        need = min(za, ze) < zcoarse[ch];
                    }
                    unsigned long long nm = __ballot(need);
                    while (nm) {
                        const int j = __ffsll((long long)nm) - 1;
                        nm &= nm - 1ull;
                        const int xb = (cb + j) << 6, x = xb + lane;
                        const bool in = x >= bbeg && x <= bend;
                        const int zi = wadd(bS0, wmul(x - bbeg, bI0)) >> 16;
                        const int zold = x < W ? zrow[x] : INT_MIN;
                        const bool ps = in && zold > zi;
                        const unsigned long long pm = __ballot(ps);
                        if (pm == 0ull) continue;
                        if (ps) zrow[x] = zi;
                        int m = ps ? zi : zold;
#pragma unroll
                        for (int d = 32; d > 0; d >>= 1) m = max(m, __shfl_xor(m, d));
                        if (lane == 0) zcoarse[cb + j] = m;
                        if (!immediate && nent == 64) {
                            // the chunks of this span recorded so far are written with the rest
                            long_seqs |= 1ull << seq;
                            seq++;
                            go_immediate();
                            sp = span_of(src);
                        }
                        if (!immediate) {
                            long_used |= 1ull << src;
                            if (lane == nent) {
                                ent_xb = xb; ent_tag = src | (seq << 8); ent_carry = carry;
                                ent_lo = (unsigned)pm; ent_hi = (unsigned)(pm >> 32);
                            }
                            nent++;
                        } else if (ps) {
                            crow[x] = span_colour(sp, x, carry + __popcll(pm & lt_mask));
                        }
                        carry += __popcll(pm);
                    }
                }
                if (!immediate) { long_seqs |= 1ull << seq; seq++; }
                LDS_WAVE_SYNC();
            }
        }
        pass_turn(&turn[0], k, lane);
        if (!immediate) {
            // ---- colours: nothing shared is touched, the groups behind go on meanwhile
            attrs(pass != 0u || ((long_used >> lane) & 1ull));
            colours(pass);
            // the first recorded chunks of long spans: their texel loads together, then the light
#pragma unroll
            for (int q = 0; q < LONG_PRE; q++) {
                lcol[q] = 0;
                if (q < nent) {
                    const unsigned long long pm = (unsigned long long)(unsigned)RL((int)ent_lo, q) |
                                                  ((unsigned long long)(unsigned)RL((int)ent_hi, q) << 32);
                    const Span sp = span_of(RL(ent_tag, q) & 0xff);
                    const int xb = RL(ent_xb, q);
                    if ((pm >> lane) & 1ull) {
                        const int *tp = texel_address<NA>(a, sp.h, sp.poly, xb + lane, y, sp.beg, sp.end, sp.S, sp.I);
                        lcol[q] = tp ? *tp : sp.h.z;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < LONG_PRE; q++) {
                if (q < nent) {
                    const unsigned long long pm = (unsigned long long)(unsigned)RL((int)ent_lo, q) |
                                                  ((unsigned long long)(unsigned)RL((int)ent_hi, q) << 32);
                    const Span sp = span_of(RL(ent_tag, q) & 0xff);
                    const int xb = RL(ent_xb, q), cr = RL(ent_carry, q);
                    const int bkind = (sp.h.w >> 8) & 0xff;
                    if (((pm >> lane) & 1ull) && NA > 1 && bkind != KIND_FLAT) {
                        const int kl = bkind >= KIND_TEX_LIN ? cr + __popcll(pm & lt_mask) : xb + lane - sp.beg;
                        lcol[q] = cl_light(lcol[q], wadd(sp.S[1 % NA], wmul(kl, sp.I[1 % NA])) >> 16,
                                           wadd(sp.S[2 % NA], wmul(kl, sp.I[2 % NA])) >> 16,
                                           wadd(sp.S[3 % NA], wmul(kl, sp.I[3 % NA])) >> 16);
                    }
                }
            }
            // ---- colour writes in draw order: groups by turn, inside a group by sequence number
            wait_turn(&turn[1], k);
            flush(true);
        }
        pass_turn(&turn[1], k, lane);
    }
    __syncthreads();
    for (int p = t; p < W; p += 64 * NW) {
        a.zbuf[g0 + p] = zrow[p];
        a.cbuf[g0 + p] = crow[p];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) frags += __shfl_down(frags, d);
    if (lane == 0 && frags) atomicAdd(&(a.status + (blockIdx.x % STATUS_SLOTS))->fragments, frags);
}

// One workgroup per screen tile; LDS holds the tile's packed (depth, draw index) keys.
//
// Depth pass.  Lanes take the tile's polygons one each and publish their depth-pass data
// in LDS; the rows of a wave's 64 polygons inside the tile are then numbered through a
// prefix sum and dealt one row per lane, whatever the polygon heights, and each lane walks
// its row pixel by pixel, lowering keys with ds_min_u64.  (A wave-wide sweep of large
// polygons, one pixel column per lane, was measured slower than this for every polygon
// size: it leaves most lanes idle on all but full-width spans.)  No lane waits for another
// workgroup-wide: the minimum is commutative, so there are no barriers between polygons.
//
// Colour pass.  After a barrier every pixel of the tile knows its winner.  Each lane
// takes pixels, looks up the winner's row (ends, start values, increments) and computes
// that one colour: CL_light of the base colour or texel (GraphicsPoly.cpp:372,486).
// Pixels that no polygon of this batch won keep what the frame held (or the clear
// colour).  The frame is read (depth only) at most once per tile and written once.

constexpr unsigned HIZ_MIN_REFS = 1024;                   // tile list length from which the coarse depth test is used

// NT threads a workgroup: 4 waves for a 64-pixel tile as a rule (four workgroups a CU), 8 for the
// 128-pixel tile, and 8 for the 64-pixel tile too when the batch before had tiles with lists many
// times the average (an object drawn small: a few tiles hold most of the polygons, and the fill
// lasts as long as the workgroup of the longest list; with twice the waves on it, one million
// triangles rotating at a distance fill in 107 instead of 162 us, while the evenly spread frame
// takes 84 instead of 76 us that way).
// The staging rows of the narrow tile's flat and Gouraud fills with four waves are dynamic LDS that the launch adds
// (RASTER_STAGE_DYNAMIC bytes), not a static array: with all 40 KB static the compiler sees four workgroups a CU,
// hence four waves a SIMD, and lets the kernel have 97 registers -- one more than five waves a SIMD can have.  A
// SIMD's 512 registers are what the fill's waves and the set-up waves of the batch behind them share: at 96 each,
// three of the fill and two of the fast set-up fit, at 97 and 98 (what both kernels took when left alone) four in all.
constexpr unsigned RASTER_STAGE_DYNAMIC = 256 * 5 * 16;
template <int NA, int TW, int NT>
__device__ __forceinline__ void raster_body(const RasterArgs &a)
{
    constexpr int RASTER_THREADS = NT;
    // The class of polygons that lie wholly behind their tile (below) belongs to the wide tile, which the
    // library takes for fragment-heavy batches: one million triangles of radius 128 at 1080p are 2.3 G
    // depth tests on 2 M pixels, and evaluating the rows of polygons that cannot be seen was 60 % of
    // that frame.  The narrow tile's kernels, which draw the small triangles, do without its registers.
    constexpr bool FAR_CLASS = TW == 128;
    constexpr int ZRUNS = TH * (TW / 8);      // 8-pixel runs of the coarse table, one thread each
    constexpr int ESZ = edge_ints(NA);
    // rows are KW keys apart: with a stride of exactly TW every row would start in the same
    // LDS bank, and the lanes of the depth pass walk different rows at similar columns
    constexpr int KW = TW + 1;
    __shared__ long long keys[TH * KW];
    // Coarse depth test for tiles with long lists (deep overdraw): the largest depth held in
    // each run of 8 pixels of a tile row, refreshed from the keys at the start of every round.
    // Keys only ever decrease, so a stale entry is merely conservative: a piece of a span
    // whose smallest depth is greater than the entry cannot lower any key there and is
    // skipped without touching the keys.  Its fragments are counted all the same.
    __shared__ int zmax[TH * (TW / 8)];
    // per wave: 64 polygons x {edge 0, edge 1, edge 2 as {x1|y1<<16, edge_pack, Zstart, Zinc}, {hr of edges 0 and 1},
    // {hr of edge 2, polygon, edges | first row << 5 | row offset << 19}}
    constexpr bool DYN = TW == 64 && NA <= 4 && NT == 256;
    __shared__ int4 stage_s[DYN ? 1 : RASTER_THREADS * 5];
    extern __shared__ int4 stage_d[];
    int4 *const stage = DYN ? stage_d : stage_s;
    __shared__ double2 rlut[64];
    // per wave: one bit per row of a round (64 polygons x TH rows), set at the last row of every polygon
    __shared__ unsigned long long marks[(RASTER_THREADS / 64) * (TH * 64 / 64)];

    const int t = threadIdx.x, lane = t & 63;
    // Workgroups start in index order and a CU serves its oldest waves first, so the tiles that
    // take longest should come first: those of the two border columns, where polygons cut by
    // the x clip have more than three edges and rows cost a memory round trip.  Measured on
    // one million small triangles at 1080p: started last, such tiles ended 12-20 us after all
    // others and set the kernel's duration.
    int tile = blockIdx.x;
    if (a.tiles_x >= 3) {
        const int nb = 2 * a.tiles_y;
        if (tile < nb) tile = (tile >> 1) * a.tiles_x + ((tile & 1) ? a.tiles_x - 1 : 0);
        else { const int q = tile - nb, w = a.tiles_x - 2; tile = (q / w) * a.tiles_x + 1 + q % w; }
    }
    const int tx0 = (tile % a.tiles_x) * TW, ty0 = (tile / a.tiles_x) * TH;
    const unsigned lbeg = a.tile_offset[tile];
    unsigned lcnt = a.tile_fill[tile];
    if ((unsigned long long)lbeg + lcnt > a.list_capacity)      // list overflow (reported by the scan): stay inside the list
        lcnt = a.list_capacity > lbeg ? (unsigned)(a.list_capacity - lbeg) : 0u;
    if (lcnt == 0 && !a.clear_pending) return;       // untouched tile of a frame that is already in HBM
    if (ty0 + TH <= a.band_y0 || ty0 >= a.band_y1) return;      // outside the band this context draws
    if (NA == 6 && (a.force_order || *a.order_flag == a.order_gen)) return;      // the draw-order kernel fills this batch

    // ---- tile in: depth only
    for (int p = t; p < TW * TH; p += RASTER_THREADS) {
        int lx = p % TW, ly = p / TW, gx = tx0 + lx, gy = ty0 + ly;
        int z = INT_MAX;
        if (!a.clear_pending && gx < a.W && gy < a.H) z = a.zbuf[(size_t)gy * a.W + gx];
        keys[ly * KW + lx] = depth_key(z, 0u);
    }
    for (int p = t; p < (RASTER_THREADS / 64) * TH; p += RASTER_THREADS) marks[p] = 0ull;
    // Every thread refreshes its entry of the coarse table at the start of each round, with no
    // barrier before other waves read it: until then the entry must not rule anything out.
    if (t < ZRUNS) zmax[t] = INT_MAX;
    fill_rcp_table(rlut, t);
    __syncthreads();

    unsigned long long frags = 0;
    const bool hiz = lcnt >= HIZ_MIN_REFS;

    // ---- depth pass.  Loads are software-pipelined: while a round's rows are drawn, the
    // header and the depth pieces of the three first edge records of the next round's
    // polygons are in flight, and so is the list entry of the round after that.  The edge
    // pieces do not wait for the header: the record of every slot exists, whatever its edge count.
    const unsigned rounds = (lcnt + RASTER_THREADS - 1) / RASTER_THREADS;
    const int4 zero4 = make_int4(0, 0, 0, 0);
    // list entries are dealt to the waves in turn (entry e of a round goes to wave e % 4),
    // so a short list or a short last round still gives every wave its share
    const unsigned slot = (unsigned)(lane * (RASTER_THREADS / 64) + (t >> 6));
    // (an entry that is not a polygon of this batch is dropped: unsigned compare)
    const unsigned npu = (unsigned)a.np;
    int poly_c = slot < lcnt ? (int)a.tile_list[lbeg + slot] : -1;
    int poly_n = RASTER_THREADS + slot < lcnt ? (int)a.tile_list[lbeg + RASTER_THREADS + slot] : -1;
    if ((unsigned)poly_c >= npu) poly_c = -1;
    if ((unsigned)poly_n >= npu) poly_n = -1;
    int4 h_c = zero4, e0_c = zero4, e1_c = zero4, e2_c = zero4;
    if (poly_c >= 0) {
        h_c = *record_header(a.edges, ESZ / 4, poly_c);
        e0_c = *edge_piece(a.edges, a.np, ESZ / 4, 0, 0, poly_c);
        e1_c = *edge_piece(a.edges, a.np, ESZ / 4, 1, 0, poly_c);
        e2_c = *edge_piece(a.edges, a.np, ESZ / 4, 2, 0, poly_c);
    }
    for (unsigned rd = 0; rd < rounds; rd++) {
        if (hiz && t < ZRUNS) {
            const long long *k8 = keys + (t / (TW / 8)) * KW + (t % (TW / 8)) * 8;
            int m = (int)(k8[0] >> 32);
#pragma unroll
            for (int q = 1; q < 8; q++) m = max(m, (int)(k8[q] >> 32));
            zmax[t] = m;
        }
        // The largest depth the whole tile holds (from the coarse table; entries that other waves
        // have not refreshed yet are older, so larger): a polygon all of whose depths are greater
        // cannot lower any key of the tile.
        int tile_far = INT_MAX;
        if (FAR_CLASS && hiz) {
            int m = INT_MIN;
#pragma unroll
            for (int q = 0; q < ZRUNS / 64; q++) m = max(m, zmax[q * 64 + lane]);
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) m = max(m, __shfl_xor(m, d));
            tile_far = m;
        }
        int4 h_n = zero4, e0_n = zero4, e1_n = zero4, e2_n = zero4;
        if (poly_n >= 0) {
            h_n = *record_header(a.edges, ESZ / 4, poly_n);
            e0_n = *edge_piece(a.edges, a.np, ESZ / 4, 0, 0, poly_n);
            e1_n = *edge_piece(a.edges, a.np, ESZ / 4, 1, 0, poly_n);
            e2_n = *edge_piece(a.edges, a.np, ESZ / 4, 2, 0, poly_n);
        }
        const unsigned li2 = (rd + 2) * RASTER_THREADS + slot;
        int poly_nn = li2 < lcnt ? (int)a.tile_list[lbeg + li2] : -1;
        if ((unsigned)poly_nn >= npu) poly_nn = -1;

        const int poly = poly_c;
        int ne = 0, r0 = 0, r1 = 0, cx0 = 0, cx1 = -1;
        Edge4 ed[3];
        ed[0].xy1 = e0_c.x; ed[0].xy2 = e0_c.y; ed[0].z0 = e0_c.z; ed[0].zi = e0_c.w;
        ed[1].xy1 = e1_c.x; ed[1].xy2 = e1_c.y; ed[1].z0 = e1_c.z; ed[1].zi = e1_c.w;
        ed[2].xy1 = e2_c.x; ed[2].xy2 = e2_c.y; ed[2].z0 = e2_c.z; ed[2].zi = e2_c.w;
        if (poly >= 0) {
            ne = h_c.w & 0xff;
            r0 = max(h_c.x & 0xffff, ty0);
            r1 = min((h_c.x >> 16) & 0xffff, ty0 + TH);
            cx0 = max(h_c.y & 0xffff, tx0);
            cx1 = min((h_c.y >> 16) & 0xffff, tx0 + TW - 1);
        }
        poly_c = poly_n; poly_n = poly_nn;
        h_c = h_n; e0_c = e0_n; e1_c = e1_n; e2_c = e2_n;
        const bool live = poly >= 0 && r0 < r1 && cx0 <= cx1;

        // the rows of this round's polygons inside the tile are numbered through a wave-wide
        // prefix sum and dealt one row per lane, 64 rows per step, so lanes stay busy
        // whatever the mix of polygon heights.  The live polygons publish their depth-pass
        // data in the wave's own LDS staging rows, in lane order without gaps, and set the
        // bit of their last row in the wave's row marks; the lane that draws row r finds
        // its owner by counting the polygons that end before r -- the marks below r -- and
        // reads the owner's data back (ordering inside one wave: LDS operations complete
        // in issue order).
        {
            const int cnt = live ? r1 - r0 : 0;
            const bool owns = cnt > 0;
            // The wave's polygons are taken in the order of their width inside the tile (a
            // counting sort over 16 width classes, ballots only): a step of 64 rows lasts as
            // long as its longest span, and rows of polygons of similar width have spans of
            // similar length.  `rank` is a polygon's place in that order (polygons without
            // rows come last), and rows are numbered in that order.
            const int wdt = cx1 - cx0 + 1;
            // A polygon of up to three edges that lies wholly behind the tile's largest depth: its
            // smallest depth is at an end of one of its edges (depth runs monotonically along an edge,
            // and along a span between two edge values), provided that nothing wraps: every edge's last
            // value fits 32 bits and the values lie less than 2^31 apart, so that no difference of two of
            // them wraps either.  Such a polygon changes no key, whatever its slot: its rows are only
            // measured for the fragment count, in steps of their own (class 0, first in the order).
            bool far = false;
            if (FAR_CLASS && hiz && owns && ne <= 3 && tile_far != INT_MAX) {
                long long lo = LLONG_MAX, hi = LLONG_MIN;
                bool fits = true;
#pragma unroll
                for (int e = 0; e < 3; e++) {
                    if (e < ne) {
                        const int Pe = ed[e].xy2;
                        const int L = max(Pe & 0x7fff, (Pe >> 15) & 0x7fff);
                        const long long za = (long long)ed[e].z0, zb = za + (long long)L * (long long)ed[e].zi;
                        fits = fits && zb == (long long)(int)zb;
                        lo = min(lo, min(za, zb));
                        hi = max(hi, max(za, zb));
                    }
                }
                far = fits && hi - lo < (1ll << 31) && (int)(lo >> 16) > tile_far;
            }
            // polygons with more than three edges take the last class: their rows fetch edges from
            // memory, and together at the end of the order they make few steps pay for that path
            const int cls = !owns ? 17 : (far ? 0 : 1 + (ne > 3 ? 15 : (wdt <= 16 ? (wdt - 1) >> 1 : 8 + min((wdt - 17) >> 3, 7))));
            int rank = 0, placed = 0;
#pragma unroll
            for (int b = FAR_CLASS ? 0 : 1; b <= 17; b++) {
                const unsigned long long m = __ballot(cls == b);
                const int within = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                rank = cls == b ? placed + within : rank;
                placed += __popcll(m);
            }
            // row counts to their places, prefix sum there, each polygon fetches its own
            int incl_s = __builtin_amdgcn_ds_permute(rank << 2, cnt);
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(incl_s, d);
                if (lane >= d) incl_s += o;
            }
            const int total = __shfl(incl_s, 63);
            const int incl = __builtin_amdgcn_ds_bpermute(rank << 2, incl_s);
            int4 *st = stage + (size_t)(t & ~63) * 5;       // this wave's 64 x 5 pieces
            unsigned long long *mk = marks + (t >> 6) * TH;
            const int last = incl - 1;
            if (owns) {
                // what the rows of this polygon share is made here, once: the packed deltas and
                // the reciprocal of each edge (an edge the polygon does not have gets a first
                // row that no row of the screen equals)
                int P[3], ady[3];
                double hr[3];
#pragma unroll
                for (int e = 0; e < 3; e++) {
                    P[e] = e < ne ? ed[e].xy2 : 0;             // the record keeps the packed deltas
                    ady[e] = (P[e] >> 15) & 0x7fff;
                    st[rank * 5 + e] = make_int4(e < ne ? ed[e].xy1 : (int)0xffff0000, P[e], ed[e].z0, ed[e].zi);
                }
                // the three table reads together, and one test for heights the table does not hold
                // (a tile with a long list runs alone on its CU at the end, where every dependent
                // LDS round trip of this step shows)
#pragma unroll
                for (int e = 0; e < 3; e++) hr[e] = rlut[ady[e] < 64 ? ady[e] : 63].y;
                if (__ballot(max(ady[0], max(ady[1], ady[2])) >= 64)) {
#pragma unroll
                    for (int e = 0; e < 3; e++) {
                        int big = ady[e];
                        asm volatile("" : "+v"(big));        // see rcp_pair
                        if (big >= 64) hr[e] = 0.5 * drcp(big);
                    }
                }
                st[rank * 5 + 3] = make_int4(__double2loint(hr[0]), __double2hiint(hr[0]), __double2loint(hr[1]), __double2hiint(hr[1]));
                st[rank * 5 + 4] = make_int4(__double2loint(hr[2]), __double2hiint(hr[2]), poly, ne | (far ? 16 : 0) | (r0 << 5) | ((incl - cnt) << 19));
                atomicOr(&mk[last >> 6], 1ull << (last & 63));
            }
            LDS_WAVE_SYNC();
            for (int rb = 0; rb < total; rb += 64) {
                // A SIMD issues from its oldest wave first, so of the workgroups that share a CU the
                // first to start would run at its own pace and the last would be left to finish
                // alone, one wave a SIMD, which cannot keep the SIMD busy (lifetimes of 50, 59, 70
                // and 83 us for the four measured).  Every step takes the next of the four wave
                // priorities, offset by the workgroup's place in the start order: each wave leads a
                // quarter of the time and the four end within a few microseconds of each other.
                switch ((a.start_rank_shift ? (int)(blockIdx.x >> a.start_rank_shift) + (rb >> 6) + (int)rd : 0) & 3) {
                case 0: __builtin_amdgcn_s_setprio(0); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                default: __builtin_amdgcn_s_setprio(3); break;
                }
                const int r = rb + lane;
                // polygons that end before this step's first row, then those that end inside the step before r
                const int before = __popcll(__ballot(owns && last < rb));
                const unsigned long long word = mk[rb >> 6];
                const int lo = before + __builtin_amdgcn_mbcnt_hi((unsigned)(word >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)word, 0u));
                int4 ow = zero4;
                if (r < total) ow = st[lo * 5 + 4];
                if (FAR_CLASS && hiz && __ballot(r < total && !(ow.w & 16)) == 0ull) {
                    // every row of this step belongs to a polygon behind the tile: its length counts, no more
                    if (r < total) {
                        const int y = ((ow.w >> 5) & 0x3fff) + (r - (int)((unsigned)ow.w >> 19));
                        EdgeP oe[3];
                        const int4 h01 = st[lo * 5 + 3];
#pragma unroll
                        for (int e = 0; e < 3; e++) {
                            const int2 q = *(const int2 *)&st[lo * 5 + e];
                            oe[e].xy1 = q.x; oe[e].P = q.y; oe[e].z0 = 0; oe[e].zi = 0;
                        }
                        oe[0].hr = __hiloint2double(h01.y, h01.x);
                        oe[1].hr = __hiloint2double(h01.w, h01.z);
                        oe[2].hr = __hiloint2double(ow.y, ow.x);
                        int beg, end;
                        if (extent_row3_pre(oe, y, beg, end)) {
                            const int xa = max(beg, tx0), xb = min(end, tx0 + TW - 1);
                            if (xa <= xb) frags += (unsigned long long)(xb - xa + 1);
                        }
                    }
                    continue;
                }
                if (r < total) {
                    const int opoly = ow.z, one = ow.w & 15, y = ((ow.w >> 5) & 0x3fff) + (r - (int)((unsigned)ow.w >> 19));
                    const unsigned id = (unsigned)opoly + 1u;
                    int beg, end, zs, zi;
                    bool ok;
                    if (one <= 3) {
                        EdgeP oe[3];
                        const int4 h01 = st[lo * 5 + 3];
#pragma unroll
                        for (int e = 0; e < 3; e++) {
                            const int4 q = st[lo * 5 + e];
                            oe[e].xy1 = q.x; oe[e].P = q.y; oe[e].z0 = q.z; oe[e].zi = q.w;
                        }
                        oe[0].hr = __hiloint2double(h01.y, h01.x);
                        oe[1].hr = __hiloint2double(h01.w, h01.z);
                        oe[2].hr = __hiloint2double(ow.y, ow.x);
                        ok = depth_row3_pre(oe, y, rlut, beg, end, zs, zi);
                    } else {
                        ok = depth_row_mem(a.edges, ESZ / 4, a.np, opoly, one, y, beg, end, zs, zi);
                    }
                    if (ok) {
                        const int xa = max(beg, tx0), xb = min(end, tx0 + TW - 1);
                        if (xa <= xb) {
                            frags += (unsigned long long)(xb - xa + 1);
                            long long *krow = keys + (y - ty0) * KW - tx0;
                            int z = wadd(zs, wmul(xa - beg, zi));
                            int x = xa;
                            // the coarse test needs the depth to run monotonically along the span: no 32-bit wrap
                            const long long za = (long long)zs + (long long)(xa - beg) * zi, zb = (long long)zs + (long long)(xb - beg) * zi;
                            if (hiz && za == (int)za && zb == (int)zb) {
                                const int *zrow = zmax + (y - ty0) * (TW / 8);
                                while (x <= xb) {
                                    const int bend = min(xb, x | 7);
                                    const int z1 = wadd(z, wmul(bend - x, zi));
                                    if ((min(z, z1) >> 16) > zrow[(x - tx0) >> 3]) { z = wadd(z1, zi); x = bend + 1; continue; }
                                    for (; x <= bend; x++) {
                                        atomicMin(&krow[x], depth_key(z >> 16, id));
                                        z = wadd(z, zi);
                                    }
                                }
                            }
                            for (; x + 3 <= xb; x += 4) {           // four pixels per trip: constant offsets, one loop test
#pragma unroll
                                for (int k = 0; k < 4; k++) {
                                    atomicMin(&krow[x + k], depth_key(z >> 16, id));
                                    z = wadd(z, zi);
                                }
                            }
                            for (; x <= xb; x++) {
                                atomicMin(&krow[x], depth_key(z >> 16, id));
                                z = wadd(z, zi);
                            }
                        }
                    }
                }
            }
            LDS_WAVE_SYNC();
            if (lane * 64 < total) mk[lane] = 0ull;      // the marks of this round
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();

    __builtin_amdgcn_s_setprio(0);
    // ---- colour pass + tile out: every pixel's colour from its winner (GraphicsPoly.cpp:372, 486), the frame written once.
    unsigned long long qrows = 0;
    constexpr int CG = 4;
    if (NA >= 4) {
        // Gouraud and textured fills: a winner's row is evaluated once for each run of neighbouring
        // pixels that it won, not once per pixel (on one million small triangles at 1080p a run is 2.5
        // pixels on average, on ten million textured ones at 4K 5.9, and the row's ends, owners and
        // increments cost ten times what a pixel's colour costs once they are known).  The wave takes
        // four runs of 64 pixels at a time.  The first pixel of every run of equal winners -- its leader
        // -- is numbered through ballots; the leaders' winners go to a list in the wave's staging rows;
        // 64 leaders at a time, one per lane, are evaluated into a table (Gouraud: {colour, first pixel,
        // light at the first pixel, light increments}; textured: header words, polygon, both ends, light
        // and texture coordinates at the first pixel and their increments); and every pixel whose leader
        // is in the table takes its colour from there.
        constexpr int RP = NA == 4 ? 2 : 4;                     // 16-byte pieces of a table entry
        int4 *res = stage + (size_t)(t & ~63) * 5;            // 64 leaders x RP pieces
        unsigned *lead = (unsigned *)(res + 64 * RP);           // up to 256 leaders (the wave's staging rows hold 320 pieces)
        for (int p0 = t; p0 < TW * TH; p0 += RASTER_THREADS * CG) {
            long long kk[CG];
            unsigned id[CG];
            int off[CG + 1], pl[CG];
            off[0] = 0;
#pragma unroll
            for (int g = 0; g < CG; g++) {
                const int p = p0 + g * RASTER_THREADS;
                const int lx = p % TW, ly = p / TW, gx = tx0 + lx, gy = ty0 + ly;
                const bool on = p < TW * TH && gx < a.W && gy < a.H;
                kk[g] = on ? keys[ly * KW + lx] : 0ll;
                id[g] = (unsigned)kk[g];
                if (on && id[g] == 0u && a.clear_pending) {
                    const size_t gi = (size_t)gy * a.W + gx;
                    a.zbuf[gi] = INT_MAX; a.cbuf[gi] = a.clear_colour;
                }
                const unsigned left = (unsigned)__shfl_up((int)id[g], 1);
                const bool leads = id[g] != 0u && (lane == 0 || left != id[g]);
                const unsigned long long m = __ballot(leads);
                const int mine = off[g] + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                if (leads) lead[mine] = id[g];
                // the leader of this pixel's run: the last one at or before this lane
                pl[g] = mine - (leads ? 0 : 1);
                off[g + 1] = off[g] + __popcll(m);
            }
            const int total = off[CG];
            LDS_WAVE_SYNC();
            for (int base = 0; base < total; base += 64) {
                const int i = base + lane;
                if (i < total) {
                    const int g = (i >= off[1] ? 1 : 0) + (i >= off[2] ? 1 : 0) + (i >= off[3] ? 1 : 0);
                    const int gy = ty0 + (p0 + g * RASTER_THREADS) / TW;       // the same for every lane of the wave
                    const int poly = (int)(lead[i] - 1u);
                    const int4 h = *record_header(a.edges, ESZ / 4, poly);
                    int2 xy[3];
#pragma unroll
                    for (int e = 0; e < 3; e++) xy[e] = *(const int2 *)edge_piece(a.edges, a.np, ESZ / 4, e, 0, poly);
                    const int ne = h.w & 0xff, kind = (h.w >> 8) & 0xff;
                    int S[NA], E[NA], I[NA], beg = 0, end = 0;
#pragma unroll
                    for (int q = 0; q < NA; q++) { S[q] = 0; I[q] = 0; }
                    if (kind != KIND_FLAT) {
                        row_ends<NA>(a.edges, a.np, poly, ne, gy, beg, end, S, E, xy, rlut);
                        int span = end - beg;
                        if (span == 0) span = 1;
                        const double2 rs = rcp_pair(rlut, span);
#pragma unroll
                        for (int q = 1; q < NA; q++) I[q] = tdivp(wsub(E[q], S[q]), rs);
                    }
                    if (NA == 4) {
                        // bit 16 of the second word: the colour as it is (flat)
                        res[lane * RP] = make_int4(h.z, kind != KIND_FLAT ? beg : 1 << 16, S[1 % NA], S[2 % NA]);
                        res[lane * RP + 1] = make_int4(S[3 % NA], I[1 % NA], I[2 % NA], I[3 % NA]);
                    } else {
                        res[lane * RP] = make_int4(h.z, h.w, poly, beg);
                        res[lane * RP + 1] = make_int4(end, S[1 % NA], S[2 % NA], S[3 % NA]);
                        res[lane * RP + 2 % RP] = make_int4(S[4 % NA], S[5 % NA], I[1 % NA], I[2 % NA]);
                        res[lane * RP + 3 % RP] = make_int4(I[3 % NA], I[4 % NA], I[5 % NA], 0);
                    }
                }
                LDS_WAVE_SYNC();
#pragma unroll
                for (int g = 0; g < CG; g++) {
                    const int j = pl[g] - base;
                    if (id[g] != 0u && (unsigned)j < 64u) {
                        const int p = p0 + g * RASTER_THREADS;
                        const int gx = tx0 + p % TW, gy = ty0 + p / TW;
                        const int4 r0 = res[j * RP], r1 = res[j * RP + 1];
                        int c = r0.x;
                        if (NA == 4) {
                            const int kl = gx - (r0.y & 0xffff);
                            if (!(r0.y >> 16))
                                c = cl_light(c, wadd(r0.z, wmul(kl, r1.y)) >> 16, wadd(r0.w, wmul(kl, r1.z)) >> 16, wadd(r1.x, wmul(kl, r1.w)) >> 16);
                        } else {
                            const int4 r2 = res[j * RP + 2 % RP], r3 = res[j * RP + 3 % RP];
                            const int kind = (r0.y >> 8) & 0xff;
                            if (kind != KIND_FLAT) {
                                int S[NA], I[NA];
                                S[0] = 0; I[0] = 0;
                                S[1 % NA] = r1.y; S[2 % NA] = r1.z; S[3 % NA] = r1.w; S[4 % NA] = r2.x; S[5 % NA] = r2.y;
                                I[1 % NA] = r2.z; I[2 % NA] = r2.w; I[3 % NA] = r3.x; I[4 % NA] = r3.y; I[5 % NA] = r3.z;
                                if (kind >= KIND_TEX_LIN && ((r2.z | r2.w | r3.x) != 0)) qrows++;
                                c = pixel_colour<NA>(a, make_int4(0, 0, r0.x, r0.y), r0.z, gx, gy, r0.w, r1.x, S, I, gx - r0.w);
                            }
                        }
                        const size_t gi = (size_t)gy * a.W + gx;
                        a.zbuf[gi] = (int)(kk[g] >> 32);
                        a.cbuf[gi] = c;
                    }
                }
                LDS_WAVE_SYNC();        // the table is rewritten by the next 64 leaders, the list by the next four runs
            }
        }
    } else
    // flat fill: a pixel's colour is its winner's, already lit (four winners' headers in flight per lane)
    for (int p0 = t; p0 < TW * TH; p0 += RASTER_THREADS * CG) {
        long long kk[CG];
        int4 hh[CG];
#pragma unroll
        for (int g = 0; g < CG; g++) {
            const int p = p0 + g * RASTER_THREADS;
            kk[g] = p < TW * TH ? keys[(p / TW) * KW + p % TW] : 0ll;
            hh[g] = zero4;
        }
#pragma unroll
        for (int g = 0; g < CG; g++) {
            const unsigned id = (unsigned)kk[g];
            if (id != 0u) hh[g] = *record_header(a.edges, ESZ / 4, (int)(id - 1u));
        }
#pragma unroll
        for (int g = 0; g < CG; g++) {
            const int p = p0 + g * RASTER_THREADS;
            if (p >= TW * TH) continue;
            const int lx = p % TW, ly = p / TW, gx = tx0 + lx, gy = ty0 + ly;
            if (gx >= a.W || gy >= a.H) continue;
            const long long k = kk[g];
            const size_t gi = (size_t)gy * a.W + gx;
            if ((unsigned)k == 0u) {
                if (a.clear_pending) { a.zbuf[gi] = INT_MAX; a.cbuf[gi] = a.clear_colour; }
                continue;
            }
            a.zbuf[gi] = (int)(k >> 32);
            a.cbuf[gi] = hh[g].z;
        }
    }

    // counters: one atomic per wave
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        frags += __shfl_down(frags, d);
        qrows += __shfl_down(qrows, d);
    }
    if (lane == 0) {
        Status *st = a.status + (blockIdx.x % STATUS_SLOTS);
        if (frags) atomicAdd(&st->fragments, frags);
        if (qrows) atomicAdd(&st->order_dependent_rows, qrows);
    }
}

template <int NA, int TW, int NT>
__global__ void __launch_bounds__(NT) raster_kernel(RasterArgs a) { raster_body<NA, TW, NT>(a); }
template <>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) raster_kernel<1, 64, 256>(RasterArgs a) { raster_body<1, 64, 256>(a); }
template <>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) raster_kernel<4, 64, 256>(RasterArgs a) { raster_body<4, 64, 256>(a); }

// G_clear (GraphicsBase.cpp:99-111) when a frame is read back with no draw after it.
__global__ void __launch_bounds__(256) fill_kernel(int4 *__restrict__ c, int4 *__restrict__ z, size_t n4, int colour)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        c[i] = make_int4(colour, colour, colour, colour);
        z[i] = make_int4(INT_MAX, INT_MAX, INT_MAX, INT_MAX);
    }
}

__global__ void fill_tail_kernel(int *c, int *z, size_t from, size_t n, int colour)
{
    size_t i = from + threadIdx.x;
    if (i < n) { c[i] = colour; z[i] = INT_MAX; }
}

// ------------------------------------------------------------------ launch wrappers

void launch_transform(hipStream_t s, const int *m12, const int *from, int *to, int n)
{
    if (n <= 0) return;
    Mat12 M;
    for (int i = 0; i < 12; i++) M.m[i] = m12[i];
    hipLaunchKernelGGL(transform_kernel, dim3((n + 255) / 256), dim3(256), 0, s, M, (const int4 *)from, to, n);
}

// Rows of the screen that a polygon of an object can touch, for routing polygons to the
// GPUs of a sort-first split before any set-up: rows[p] = {first row, last row}, or {1, 0}
// for a polygon that draws nothing.  Where every vertex lies beyond the near plane this is
// exact arithmetic on the projected vertices -- the volume test (Clipp3D.cpp:46-58), the
// projection (T_perspective, TransBs.cpp:425-432) and the facing test (T_normal_z_negative,
// TransLn.cpp:218-223) as M_render_polygon makes them (EngBase.cpp:433-470); x and y clipping
// only take rows away.  A polygon that meets the near plane, or whose data the set-up kernels
// will have to report, gets every row.
__global__ void __launch_bounds__(256) polygon_rows_kernel(SetupArgs a, int2 *__restrict__ rows)
{
    const int poly = blockIdx.x * blockDim.x + threadIdx.x;
    if (poly >= a.np) return;
    const int2 all = make_int2(0, a.H - 1), none = make_int2(1, 0);
    const int4 pd = ((const int4 *)a.polys)[poly];
    const int e = pd.z, first = pd.w;
    if (e < 1 || e > a.max_edges || first < 0 || (long long)first + e + 1 > (long long)a.npv) { rows[poly] = all; return; }
    int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
    // tuples 0 .. e (the closing one repeats the first): view space
    for (int i = 0; i <= e; i++) {
        const int vi = a.polyverts[(size_t)(first + i) * 6];
        if ((unsigned)vi >= (unsigned)a.nv) { rows[poly] = all; return; }
        const int4 p = load_vertex(a, vi);
        mn[0] = min(mn[0], p.x); mx[0] = max(mx[0], p.x);
        mn[1] = min(mn[1], p.y); mx[1] = max(mx[1], p.y);
        mn[2] = min(mn[2], p.z); mx[2] = max(mx[2], p.z);
    }
    if ((mx[2] < mn[0]) || (mx[2] < mn[1]) || (mx[2] < -mx[0]) || (mx[2] < -mx[1]) || (mx[2] <= 10)) { rows[poly] = none; return; }
    if (mn[2] < 10) { rows[poly] = all; return; }
    const int xc = a.W / 2, yc = a.H / 2;
    int lo = INT_MAX, hi = INT_MIN, px[3] = {0, 0, 0}, py[3] = {0, 0, 0};
    for (int i = 0; i <= e; i++) {
        const int4 p = load_vertex(a, a.polyverts[(size_t)(first + i) * 6]);
        const double rz = drcp(p.z);
        const int x = wadd(persp_div(p.x, a.log_focus, p.z, rz), xc), y = wadd(persp_div(p.y, a.log_focus, p.z, rz), yc);
        if (i < 3) { px[i] = x; py[i] = y; }
        lo = min(lo, y); hi = max(hi, y);
    }
    if (e >= 2) {
        const long long c = (long long)wsub(px[1], px[0]) * (long long)wsub(py[2], py[0]) -
                            (long long)wsub(py[1], py[0]) * (long long)wsub(px[2], px[0]);
        if (!(c < 0)) { rows[poly] = none; return; }
    }
    lo = max(lo, 0); hi = min(hi, a.H - 1);
    rows[poly] = lo <= hi ? make_int2(lo, hi) : none;
}

void launch_polygon_rows(hipStream_t s, const SetupArgs &a, int *rows)
{
    if (a.np <= 0) return;
    hipLaunchKernelGGL(polygon_rows_kernel, dim3((a.np + 255) / 256), dim3(256), 0, s, a, (int2 *)rows);
}

void launch_shade_points(hipStream_t s, const ShadeArgs &a, const int *points, const int *normals, int n, int *rgb)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(shade_points_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, (const int4 *)points, (const int4 *)normals,
                       n, (int4 *)rgb);
}

void launch_shade(hipStream_t s, const ShadeArgs &a)
{
    if (a.np <= 0) return;
    hipLaunchKernelGGL(shade_kernel, dim3((a.np + 255) / 256), dim3(256), 0, s, a);
}

void launch_plane_normals(hipStream_t s, const int *verts, int nv, const int *polys, int np, const int *polyverts,
                          int npv, int *out, Status *status)
{
    if (np <= 0) return;
    hipLaunchKernelGGL(plane_normals_kernel, dim3((np + 255) / 256), dim3(256), 0, s, verts, nv, polys, np, polyverts, npv,
                       (int4 *)out, status);
}

void launch_setup(hipStream_t s, int dim, bool screen, const SetupArgs &a)
{
    if (a.np <= 0) return;
    dim3 g((a.np + SETUP_THREADS - 1) / SETUP_THREADS), b(SETUP_THREADS);
    if (screen || !a.queue) {
        // every polygon through the general kernel
        if (screen) {
            if (dim == 3) hipLaunchKernelGGL((setup_kernel<3, true>), g, b, 0, s, a);
            else if (dim == 6) hipLaunchKernelGGL((setup_kernel<6, true>), g, b, 0, s, a);
            else hipLaunchKernelGGL((setup_kernel<8, true>), g, b, 0, s, a);
        } else {
            if (dim == 3) hipLaunchKernelGGL((setup_kernel<3, false>), g, b, 0, s, a);
            else if (dim == 6) hipLaunchKernelGGL((setup_kernel<6, false>), g, b, 0, s, a);
            else hipLaunchKernelGGL((setup_kernel<8, false>), g, b, 0, s, a);
        }
        return;
    }
    // fast kernel for unclipped triangles; it queues the rest: triangles that only meet a
    // screen border for the screen-mode kernel (projected tuples), everything else for the
    // general kernel
    dim3 gf((a.np + SETUP_FAST_THREADS - 1) / SETUP_FAST_THREADS), bf(SETUP_FAST_THREADS);
    g = dim3(std::min((a.queue_cap + SETUP_THREADS - 1) / SETUP_THREADS, 8u), 2 * QUEUES);
    if (a.objects) {
        if (dim == 3) hipLaunchKernelGGL((setup_fast_kernel<3, true>), gf, bf, 0, s, a);
        else if (dim == 6) hipLaunchKernelGGL((setup_fast_kernel<6, true>), gf, bf, 0, s, a);
        else hipLaunchKernelGGL((setup_fast_kernel<8, true>), gf, bf, 0, s, a);
        if (dim == 3) hipLaunchKernelGGL((setup_queues_objects_kernel<3>), g, b, 0, s, a);
        else if (dim == 6) hipLaunchKernelGGL((setup_queues_objects_kernel<6>), g, b, 0, s, a);
        else hipLaunchKernelGGL((setup_queues_objects_kernel<8>), g, b, 0, s, a);
        return;
    }
    if (dim == 3) hipLaunchKernelGGL((setup_fast_kernel<3, false>), gf, bf, 0, s, a);
    else if (dim == 6) hipLaunchKernelGGL((setup_fast_kernel<6, false>), gf, bf, 0, s, a);
    else hipLaunchKernelGGL((setup_fast_kernel<8, false>), gf, bf, 0, s, a);
    if (dim == 3) hipLaunchKernelGGL((setup_queues_kernel<3>), g, b, 0, s, a);
    else if (dim == 6) hipLaunchKernelGGL((setup_queues_kernel<6>), g, b, 0, s, a);
    else hipLaunchKernelGGL((setup_queues_kernel<8>), g, b, 0, s, a);
}

void launch_binning(hipStream_t s, const PrimHdr *hdr, int np, int twl, int tiles_x, int ntiles, unsigned *count,
                    unsigned *offset, unsigned *cursor, unsigned *list, unsigned long long capacity,
                    unsigned *queue_count, Status *status, hipEvent_t after_scan, bool counted, hipEvent_t done,
                    const int *order_flag, int order_gen, int order_force)
{
    const bool lds = ntiles <= BIN_MAX_TILES;
    if (np > 0 && !counted) {
        if (lds && np >= BIN_LARGE_BATCH) hipLaunchKernelGGL((count_kernel<16>), dim3((np + BIN_THREADS * 16 - 1) / (BIN_THREADS * 16)), dim3(BIN_THREADS), (size_t)ntiles * sizeof(unsigned), s, hdr, np, twl, tiles_x, ntiles, count, order_flag, order_gen, order_force);
        else if (lds) hipLaunchKernelGGL((count_kernel<8>), dim3((np + BIN_THREADS * 8 - 1) / (BIN_THREADS * 8)), dim3(BIN_THREADS), (size_t)ntiles * sizeof(unsigned), s, hdr, np, twl, tiles_x, ntiles, count, order_flag, order_gen, order_force);
        else hipLaunchKernelGGL(count_direct_kernel, dim3((np + 255) / 256), dim3(256), 0, s, hdr, np, twl, tiles_x, count);
    }
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, count, offset, cursor, ntiles, capacity, queue_count, status);
    if (after_scan) (void)hipEventRecord(after_scan, s);
    if (np > 0) {
        // `done` (may be null) is the scatter's own completion (see launch_raster)
        if (lds && np >= BIN_LARGE_BATCH) hipExtLaunchKernelGGL((scatter_kernel<16>), dim3((np + BIN_THREADS * 16 - 1) / (BIN_THREADS * 16)), dim3(BIN_THREADS), (size_t)ntiles * sizeof(unsigned), s, nullptr, done, 0, hdr, np, twl, tiles_x, ntiles, (const unsigned *)offset, cursor, list, capacity, order_flag, order_gen, order_force);
        else if (lds) hipExtLaunchKernelGGL((scatter_kernel<8>), dim3((np + BIN_THREADS * 8 - 1) / (BIN_THREADS * 8)), dim3(BIN_THREADS), (size_t)ntiles * sizeof(unsigned), s, nullptr, done, 0, hdr, np, twl, tiles_x, ntiles, (const unsigned *)offset, cursor, list, capacity, order_flag, order_gen, order_force);
        else hipExtLaunchKernelGGL(scatter_direct_kernel, dim3((np + 255) / 256), dim3(256), 0, s, nullptr, done, 0, hdr, np, twl, tiles_x, (const unsigned *)offset, cursor, list, capacity);
    }
}

// `start` / `stop` (either may be null) ride on the kernel's own dispatch: they take its
// start and completion without a marker of their own in the stream, which would hold the
// stream for a few microseconds each.
// `extra_lds`: dynamic LDS the launch reserves without using it (FILL_OVERLAP_LDS), which lowers
// the number of fill workgroups a CU holds.  `waves`: 4 or 8 waves a workgroup of the 64-pixel tile.
void launch_raster(hipStream_t s, int na, int tile_w, const RasterArgs &a_in, hipEvent_t start, hipEvent_t stop, unsigned extra_lds, int waves)
{
    // log2 of the CU count (rounded down): workgroup index >> that is the workgroup's place among
    // those that start on one CU, near enough for the priority rotation of the depth pass
    static std::atomic<int> cu_shift[MAX_DEVICES];
    RasterArgs a = a_in;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < MAX_DEVICES) {
        int sh = cu_shift[dev].load(std::memory_order_relaxed);
        if (sh == 0) {
            int ncu = 0;
            if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 2) ncu = 2;
            sh = 31 - __builtin_clz((unsigned)ncu);
            cu_shift[dev].store(sh, std::memory_order_relaxed);
        }
        a.start_rank_shift = sh;
    }
    dim3 g(a.tiles_x * a.tiles_y), b(tile_w == 128 || waves == 8 ? 512 : 256);
    if (tile_w == 128) {
        if (na == 1) hipExtLaunchKernelGGL((raster_kernel<1, 128, 512>), g, b, 0, s, start, stop, 0, a);
        else if (na == 4) hipExtLaunchKernelGGL((raster_kernel<4, 128, 512>), g, b, 0, s, start, stop, 0, a);
        else hipExtLaunchKernelGGL((raster_kernel<6, 128, 512>), g, b, 0, s, start, stop, 0, a);
    } else if (waves == 8) {
        if (na == 1) hipExtLaunchKernelGGL((raster_kernel<1, 64, 512>), g, b, 0, s, start, stop, 0, a);
        else if (na == 4) hipExtLaunchKernelGGL((raster_kernel<4, 64, 512>), g, b, 0, s, start, stop, 0, a);
        else hipExtLaunchKernelGGL((raster_kernel<6, 64, 512>), g, b, 0, s, start, stop, 0, a);
    } else {
        if (na == 1) hipExtLaunchKernelGGL((raster_kernel<1, 64, 256>), g, b, RASTER_STAGE_DYNAMIC + extra_lds, s, start, stop, 0, a);
        else if (na == 4) hipExtLaunchKernelGGL((raster_kernel<4, 64, 256>), g, b, RASTER_STAGE_DYNAMIC + extra_lds, s, start, stop, 0, a);
        else hipExtLaunchKernelGGL((raster_kernel<6, 64, 256>), g, b, extra_lds, s, start, stop, 0, a);
    }
}

int ordered_band_rows(int W, int H)
{
    int r = (120 * 1024) / (8 * W);
    if (r < 1) r = 1;
    if (r > 16) r = 16;
    if (r > H) r = H;
    return r;
}

// Rows per band (the granularity of the lists), cell size and LDS bytes of the list-driven
// draw-order kernel: one row's depth and colour, the row's cells and the two turn words.
void ordered_list_layout(int W, int H, int *R, int *cshift, size_t *lds)
{
    *cshift = W > 8192 ? 4 : 3;
    // bands of about 16K pixels: few enough for the bit matrix of the binning (MAX_BANDS), and most
    // polygons of a band's list have a span on each of its rows
    int r = (16 * 1024) / W;
    if (r < 1) r = 1;
    if (r > 16) r = 16;
    if (r > H) r = H;
    *R = r;
    *lds = (size_t)W * 8 + ((size_t)(W >> *cshift) + 1 + 2 + (size_t)((W + 63) >> 6)) * sizeof(unsigned);
}

bool band_lists_possible(int W, int H, int *R, int *nb)
{
    int cs;
    size_t lds;
    ordered_list_layout(W, H, R, &cs, &lds);
    *nb = (H + *R - 1) / *R;
    return *nb <= MAX_BANDS;
}

// the dynamic-LDS attribute belongs to the function on one device: once per device (which is current)
template <typename F>
static hipError_t allow_large_lds(F fn, std::atomic<bool> (&done)[MAX_DEVICES], int device)
{
    if (device < 0 || device >= MAX_DEVICES) return hipErrorInvalidDevice;
    if (!done[device].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        done[device].store(true, std::memory_order_release);
    }
    return hipSuccess;
}

constexpr int ORDERED_WAVES = 4;           // waves that share one row of the list-driven draw-order kernel
constexpr int BAND_BIN_MAX_GRID = 1024;    // workgroups of the band binning (they stride over the chunks)

// Per-band lists of a batch (needs the batch's headers: after set-up, any stream)
hipError_t launch_band_lists(hipStream_t s, int np, int device, const BandArgs &bands)
{
    static std::atomic<bool> set_count[MAX_DEVICES], set_scatter[MAX_DEVICES];
    hipError_t e;
    if ((e = allow_large_lds(band_bin_kernel<false>, set_count, device)) != hipSuccess) return e;
    if ((e = allow_large_lds(band_bin_kernel<true>, set_scatter, device)) != hipSuccess) return e;
    const int nchunks = (np + BAND_CHUNK - 1) / BAND_CHUNK;
    const size_t bits = (size_t)bands.nb * BAND_PITCH * sizeof(unsigned);
    const int bin_grid = nchunks < BAND_BIN_MAX_GRID ? nchunks : BAND_BIN_MAX_GRID;
    hipLaunchKernelGGL((band_bin_kernel<false>), dim3(bin_grid), dim3(BAND_THREADS), bits, s, bands, nchunks);
    hipLaunchKernelGGL(band_scan_kernel, dim3((bands.nb + 63) / 64), dim3(1024), 0, s, bands, nchunks);
    hipLaunchKernelGGL((band_bin_kernel<true>), dim3(bin_grid), dim3(BAND_THREADS), bits, s, bands, nchunks);
    return hipSuccess;
}

// `bands` (may be null): the per-band lists made by launch_band_lists; without them every band reads every header
hipError_t launch_ordered(hipStream_t s, const RasterArgs &a, int device, const BandArgs *bands)
{
    static std::atomic<bool> set_walk[MAX_DEVICES], set_list[MAX_DEVICES];
    hipError_t e;
    if (!bands) {
        const int R = ordered_band_rows(a.W, a.H);
        const size_t lds = (size_t)R * a.W * 8;
        if ((e = allow_large_lds(ordered_kernel<6>, set_walk, device)) != hipSuccess) return e;
        hipLaunchKernelGGL((ordered_kernel<6>), dim3((a.H + R - 1) / R), dim3(256), lds, s, a, R);
        return hipSuccess;
    }
    int R, cshift;
    size_t lds;
    ordered_list_layout(a.W, a.H, &R, &cshift, &lds);
    if (R != bands->R) return hipErrorInvalidValue;
    if ((e = allow_large_lds(ordered_list_kernel<6, ORDERED_WAVES>, set_list, device)) != hipSuccess) return e;
    hipLaunchKernelGGL((ordered_list_kernel<6, ORDERED_WAVES>), dim3(((bands->nb + 7) / 8) * 8 * R), dim3(64 * ORDERED_WAVES), lds, s, a, *bands, cshift);
    return hipSuccess;
}

// ------------------------------------------------------------------ wireframe

// The two clips of G_line (GraphicsBase.cpp:191-193) are made with dimension 2: only X and Y
// of an end point that a clipper moves are computed, and the depth read through the returned
// pointer is the third word of one of the clipper's static stores (Clipp2D.cpp:99-104, 132,
// 150).  A program that draws only wireframes never writes those words: 0, which is what a
// moved end point gets here (after filled polygons the reference's value is a leftover of
// the polygon clipper's last calls).
template <int AX>
__device__ __forceinline__ bool clip_line_xy(Tup<3> &v1, Tup<3> &v2, int bmin, int bmax, bool &bad)
{
    Tup<2> p, q, o1, o2;
    p.v[0] = v1.v[0]; p.v[1] = v1.v[1]; q.v[0] = v2.v[0]; q.v[1] = v2.v[1];
    const bool swap = !(v1.v[AX] < v2.v[AX]);
    const bool m1 = swap ? v1.v[AX] >= bmax : v1.v[AX] <= bmin;
    const bool m2 = swap ? v2.v[AX] <= bmin : v2.v[AX] >= bmax;
    bool moved;
    if (!clip_line<2, AX, true>(p, q, bmin, bmax, o1, o2, moved, bad)) return false;
    v1.v[0] = o1.v[0]; v1.v[1] = o1.v[1]; v1.v[2] = m1 ? 0 : v1.v[2];
    v2.v[0] = o2.v[0]; v2.v[1] = o2.v[1]; v2.v[2] = m2 ? 0 : v2.v[2];
    return true;
}

// G_line, GraphicsBase.cpp:165-253: the points of the line from v1 to v2 (both included),
// each tested against the depth buffer, which a line does not write.  The reference's
// later line overwrites an earlier one's pixel, so every pixel keeps the largest draw index
// among the lines that pass there (`winner`, 0 = none); line_resolve_kernel turns indices
// into colours.  Points outside rows [by0, by1) belong to another context's band.
__device__ __forceinline__ void draw_line(int W, const ClipWindow cw, int by0, int by1, const int *__restrict__ zbuf,
                                          unsigned *winner, Tup<3> v1, Tup<3> v2, unsigned id, unsigned long long &tests,
                                          bool &bad, bool clipped = false)
{
    // `clipped`: the caller has clipped the line (and chosen the depth of its end points)
    if (!clipped) {
        if (!clip_line_xy<0>(v1, v2, cw.x0, cw.x1, bad)) return;
        if (!clip_line_xy<1>(v1, v2, cw.y0, cw.y1, bad)) return;
    }
    if (bad) return;
    int dx = v2.v[0] - v1.v[0], dy = v2.v[1] - v1.v[1];
    const int sx = dx < 0 ? -1 : 1, sy = dy < 0 ? -1 : 1;
    dx = dx < 0 ? -dx : dx;
    dy = dy < 0 ? -dy : dy;
    const bool xmaj = dx > dy;
    const int long_d = xmaj ? dx : dy, short_d = xmaj ? dy : dx;
    int d = 2 * short_d - long_d;
    const int add_dl = 2 * short_d, add_dh = 2 * (short_d - long_d);
    int x = v1.v[0], y = v1.v[1], cur_z = v1.v[2];
    const int inc_z = long_d != 0 ? tdiv_any(wsub(v2.v[2], cur_z), long_d) : 0;
    for (int i = 0; i <= long_d; i++) {
        if (y >= by0 && y < by1) {
            const size_t pos = (size_t)y * W + x;
            tests++;
            if (cur_z < zbuf[pos]) atomicMax(winner + pos, id);
        }
        if (d >= 0) { d += add_dh; x += sx; y += sy; }
        else { d += add_dl; x += xmaj ? sx : 0; y += xmaj ? 0 : sy; }
        cur_z = wadd(cur_z, inc_z);
    }
}

// M_render_polygon with M_WIRE (EngBase.cpp:393-431): tuples of X Y Z, volume test,
// near-plane clip, perspective, back-face test on the first three projected tuples, then
// G_line for every edge of the clipped polygon (the polygon itself is not clipped to the
// screen; each line is).  One polygon per lane, the stages chained as in setup_one; line k
// of polygon p has draw index 16 p + k + 1.
__global__ void __launch_bounds__(128) wire_kernel(SetupArgs a, const int *__restrict__ zbuf, unsigned *winner,
                                                   int *__restrict__ colours)
{
    const int poly = blockIdx.x * blockDim.x + threadIdx.x;
    const ClipWindow cw = {a.cx0, a.cy0, a.cx1, a.cy1};
    bool bad = false, drawn = false;
    unsigned long long tests = 0;
    if (poly < a.np) {
        const int4 pd = ((const int4 *)a.polys)[poly];
        const int colour = pd.x, e = pd.z, first = pd.w;
        const int4 ax = a.polyaux ? ((const int4 *)a.polyaux)[poly] : make_int4(0, 0, 0, 0);
        colours[poly] = cl_light(colour, ax.x, ax.y, ax.z);
        bool go = true;
        if (e < 2 || e + 1 > 16 || e > a.max_edges || first < 0 || (long long)first + e + 1 > (long long)a.npv) { bad = true; go = false; }
        bool zclip = false;
        if (go) {
            int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
            for (int i = 0; i <= e; i++) {
                const int idx = a.polyverts[(size_t)(first + i) * 6];
                if ((unsigned)idx >= (unsigned)a.nv) { bad = true; go = false; break; }
                const int4 p = load_vertex(a, idx);
                mn[0] = min(mn[0], p.x); mx[0] = max(mx[0], p.x);
                mn[1] = min(mn[1], p.y); mx[1] = max(mx[1], p.y);
                mn[2] = min(mn[2], p.z); mx[2] = max(mx[2], p.z);
            }
            if (go) {           // C_volume_clipping, Clipp3D.cpp:46-58
                if ((mx[2] < mn[0]) || (mx[2] < mn[1]) || (mx[2] < -mx[0]) || (mx[2] < -mx[1]) || (mx[2] <= 10)) go = false;
                else zclip = mn[2] < 10;
            }
        }
        if (go && zclip) {
            // the clipped polygon must fit the reference's arrays (16 tuples): count first,
            // because lines are drawn as the tuples appear
            int zn = 0;
            Tup<3> zprev;
            for (int it = 0; it <= e; it++) {
                const int4 p = load_vertex(a, a.polyverts[(size_t)(first + it) * 6]);
                Tup<3> t, o1, o2;
                t.v[0] = p.x; t.v[1] = p.y; t.v[2] = p.z;
                bool moved;
                if (it > 0 && clip_line<3, 2, false>(zprev, t, 10, 0, o1, o2, moved, bad)) zn += moved ? 2 : 1;
                zprev = t;
            }
            if (zn + 1 > 16 || bad) { bad = true; go = false; }
        }
        if (go) {
            const int xc = a.W / 2, yc = a.H / 2;
            Tup<3> zprev, zfirst, q0, q1, qprev;
            zprev = Tup<3>(); zfirst = zprev; q0 = zprev; q1 = zprev; qprev = zprev;
            int zn = 0;
            for (int it = 0; go && it <= e + 1; it++) {
                Tup<3> zo0 = zprev, zo1 = zprev;
                int cz = 0;
                if (it <= e) {
                    const int4 p = load_vertex(a, a.polyverts[(size_t)(first + it) * 6]);
                    Tup<3> t;
                    t.v[0] = p.x; t.v[1] = p.y; t.v[2] = p.z;
                    if (!zclip) { zo0 = t; cz = 1; }
                    else if (it > 0) {
                        // C_polygon_z_clipping, Clipp3D.cpp:137-155
                        Tup<3> o1, o2;
                        bool moved;
                        if (clip_line<3, 2, false>(zprev, t, 10, 0, o1, o2, moved, bad)) {
                            zo0 = sel(moved, o1, o2);
                            zo1 = o2;
                            cz = moved ? 2 : 1;
                        }
                    }
                    zprev = t;
                } else if (zclip && zn > 0) {
                    zo0 = zfirst; cz = 1;       // closing tuple = first tuple, Clipp3D.cpp:156
                }
                for (int k = 0; go && k < cz; k++) {
                    const Tup<3> p = sel(k != 0, zo1, zo0);
                    if (zn == 0) zfirst = p;
                    // T_perspective, TransBs.cpp:425-432
                    const int z = p.v[2];
                    const double rz = drcp(z);
                    Tup<3> q;
                    q.v[0] = wadd(persp_div(p.v[0], a.log_focus, z, rz), xc);
                    q.v[1] = wadd(persp_div(p.v[1], a.log_focus, z, rz), yc);
                    q.v[2] = tdivr(-0x7ffff, rz);
                    const unsigned id = (unsigned)poly * 16u + (unsigned)zn;        // index of line (zn-1, zn), plus one
                    if (zn == 0) q0 = q;
                    else if (zn == 1) q1 = q;
                    else if (zn == 2) {
                        // T_normal_z_negative, TransLn.cpp:218-223
                        const long long c = (long long)wsub(q1.v[0], q0.v[0]) * (long long)wsub(q.v[1], q0.v[1]) -
                                            (long long)wsub(q1.v[1], q0.v[1]) * (long long)wsub(q.v[0], q0.v[0]);
                        if (!(c < 0)) go = false;
                        else {
                            drawn = true;
                            draw_line(a.W, cw, a.band_y0, a.band_y1, zbuf, winner, q0, q1, id - 1u, tests, bad);
                            draw_line(a.W, cw, a.band_y0, a.band_y1, zbuf, winner, q1, q, id, tests, bad);
                        }
                    } else {
                        draw_line(a.W, cw, a.band_y0, a.band_y1, zbuf, winner, qprev, q, id, tests, bad);
                    }
                    qprev = q;
                    zn++;
                }
            }
        }
    }
    Status *st = a.status + (blockIdx.x % STATUS_SLOTS);
    if (tests) atomicAdd(&st->fragments, tests);
    unsigned long long m = __ballot(drawn);
    if (m && (threadIdx.x & 63) == (__ffsll((long long)m) - 1)) atomicAdd(&st->polys_drawn, (unsigned long long)__popcll(m));
    m = __ballot(bad);
    if (m && (threadIdx.x & 63) == (__ffsll((long long)m) - 1)) atomicAdd(&st->overflow_polys, (unsigned long long)__popcll(m));
}

// G_line for n lines in draw order: lines[i] = {x1, y1, z1, x2, y2, z2, lit colour, kind}; kind 1 and
// 2 are G_dot and G_pixel at (x1, y1, z1), kind 3 a line whose ends are final (clipped by the caller)
__global__ void __launch_bounds__(128) lines_kernel(const int *__restrict__ lines, int n, int W, int H, ClipWindow cw, int by0, int by1,
                                                    const int *__restrict__ zbuf, unsigned *winner, Status *status)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long tests = 0;
    bool bad = false;
    if (i < n) {
        const int4 lo = ((const int4 *)lines)[2 * (size_t)i], hi = ((const int4 *)lines)[2 * (size_t)i + 1];
        Tup<3> v1, v2;
        v1.v[0] = lo.x; v1.v[1] = lo.y; v1.v[2] = lo.z;
        v2.v[0] = lo.w; v2.v[1] = hi.x; v2.v[2] = hi.y;
        const int kind = hi.w;
        if (kind == 0) {
            draw_line(W, cw, by0, by1, zbuf, winner, v1, v2, (unsigned)i + 1u, tests, bad);
        } else if (kind == 3) {
            // a line that its caller has clipped: both ends must be on the screen
            if ((unsigned)lo.x < (unsigned)W && (unsigned)lo.y < (unsigned)H && (unsigned)lo.w < (unsigned)W && (unsigned)hi.x < (unsigned)H)
                draw_line(W, cw, by0, by1, zbuf, winner, v1, v2, (unsigned)i + 1u, tests, bad, true);
            else bad = true;
        } else if ((unsigned)lo.x < (unsigned)W && (unsigned)lo.y < (unsigned)H && lo.y >= by0 && lo.y < by1) {
            // G_dot (GraphicsBase.cpp:133-153): on the screen and nearer than the depth buffer;
            // G_pixel (:117-123): no test at all (a pixel off the screen is dropped here, the
            // reference would write outside its buffer)
            const size_t pos = (size_t)lo.y * W + lo.x;
            if (kind == 1) tests++;
            if (kind == 2 || lo.z < zbuf[pos]) atomicMax(winner + pos, (unsigned)i + 1u);
        }
    }
    Status *st = status + (blockIdx.x % STATUS_SLOTS);
    if (tests) atomicAdd(&st->fragments, tests);
    const unsigned long long m = __ballot(bad);
    if (m && (threadIdx.x & 63) == (__ffsll((long long)m) - 1)) atomicAdd(&st->overflow_polys, (unsigned long long)__popcll(m));
}

// colour of pixel p = colour of its winning line: col[((winner - 1) >> shift) * stride + offset]
__global__ void __launch_bounds__(256) line_resolve_kernel(unsigned *winner, int *__restrict__ cbuf, size_t n,
                                                           const int *__restrict__ col, int shift, int stride, int offset)
{
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; p < n; p += step) {
        const unsigned w = winner[p];
        if (w) {
            cbuf[p] = col[(size_t)((w - 1u) >> shift) * stride + offset];
            winner[p] = 0u;
        }
    }
}

void launch_wire(hipStream_t s, const SetupArgs &a, const int *zbuf, unsigned *winner, int *colours, int *cbuf)
{
    if (a.np <= 0) return;
    hipLaunchKernelGGL(wire_kernel, dim3((a.np + 127) / 128), dim3(128), 0, s, a, zbuf, winner, colours);
    const size_t n = (size_t)a.W * a.H;
    hipLaunchKernelGGL(line_resolve_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, s, winner,
                       cbuf, n, (const int *)colours, 4, 1, 0);
}

void launch_lines(hipStream_t s, const int *lines, int n, int W, int H, int by0, int by1, ClipWindow cw, const int *zbuf,
                  unsigned *winner, int *cbuf, Status *status)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(lines_kernel, dim3((n + 127) / 128), dim3(128), 0, s, lines, n, W, H, cw, by0, by1, zbuf, winner, status);
    const size_t np = (size_t)W * H;
    hipLaunchKernelGGL(line_resolve_kernel, dim3((unsigned)std::min<size_t>((np + 255) / 256, 8192)), dim3(256), 0, s, winner,
                       cbuf, np, lines, 0, 8, 6);
}

__global__ void __launch_bounds__(256) fill_scalar_kernel(int *c, int *z, size_t n, int colour)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) { c[i] = colour; z[i] = INT_MAX; }
}

void launch_fill(hipStream_t s, int *cbuf, int *zbuf, size_t n, int colour)
{
    // 16-byte stores between a few leading and trailing words (a band may start at any row)
    const size_t lead_c = ((16 - ((uintptr_t)cbuf & 15)) & 15) / 4, lead_z = ((16 - ((uintptr_t)zbuf & 15)) & 15) / 4;
    if (lead_c != lead_z || n < 8) {
        if (n) hipLaunchKernelGGL(fill_scalar_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, s, cbuf, zbuf, n, colour);
        return;
    }
    if (lead_c) hipLaunchKernelGGL(fill_tail_kernel, dim3(1), dim3(4), 0, s, cbuf, zbuf, (size_t)0, lead_c, colour);
    cbuf += lead_c; zbuf += lead_c; n -= lead_c;
    size_t n4 = n / 4;
    if (n4) {
        size_t blocks = (n4 + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (int4 *)cbuf, (int4 *)zbuf, n4, colour);
    }
    if (n4 * 4 < n) hipLaunchKernelGGL(fill_tail_kernel, dim3(1), dim3(4), 0, s, cbuf, zbuf, n4 * 4, n, colour);
}

}  // namespace lt
