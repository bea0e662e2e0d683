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
// of the order in which lanes, waves and workgroups happen to run.
#include "lt_kernels.h"
#include <limits.h>

namespace lt {

// ------------------------------------------------------------------ small helpers

__device__ __forceinline__ int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
__device__ __forceinline__ int wsub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
__device__ __forceinline__ int wmul(int a, int b) { return (int)((unsigned)a * (unsigned)b); }
__device__ __forceinline__ int wshl(int a, int s) { return (int)((unsigned)a << s); }
// truncating signed division; b != 0.  INT_MIN / -1 wraps to INT_MIN.
__device__ __forceinline__ int tdiv(int a, int b)
{
    if (b == -1) return (int)(0u - (unsigned)a);
    return a / b;
}

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

// The same transform applied to one vertex as the setup kernel reads it, so that an
// object's view-space vertices never make a round trip through HBM.
__device__ __forceinline__ int4 load_vertex(const SetupArgs &a, int idx)
{
    int4 p = ((const int4 *)a.verts)[idx];
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
    if (lo.v[AX] <= bmin) {
        Tup<DIM> m = lo, r = hi, l = lo;
        bool whereto = true;
        int guard = 0;
        while (m.v[AX] != bmin) {
            l = sel(whereto, m, l);
            r = sel(whereto, r, m);
            midpoint(m, l, r);
            whereto = m.v[AX] < bmin;
            if (++guard > 40) { bad = true; break; }
        }
        lo = m;
        moved = !swap;
    }
    if (BOTH && hi.v[AX] >= bmax) {
        Tup<DIM> l = lo, m = hi, r = hi;
        bool whereto = false;
        int guard = 0;
        while (m.v[AX] != bmax) {
            l = sel(whereto, m, l);
            r = sel(whereto, r, m);
            midpoint(m, l, r);
            whereto = m.v[AX] < bmax;
            if (++guard > 40) { bad = true; break; }
        }
        hi = m;
        moved = moved || swap;
    }
    o1 = sel(swap, hi, lo);
    o2 = sel(swap, lo, hi);
    return true;
}

// T_perspective (TransBs.cpp:425-429): (int)(((long)x << log_focus) / z).  The 64-bit
// quotient is only needed when x << log_focus leaves 32 bits.
__device__ __forceinline__ int persp_div(int x, int lf, int z)
{
    int xs = wshl(x, lf);
    if ((xs >> lf) == x) return xs / z;
    return (int)(((long long)x * ((long long)1 << lf)) / (long long)z);
}

// ------------------------------------------------------------------ setup

constexpr int KIND_FLAT = 2, KIND_SHADED = 4, KIND_TEX_LIN = 8;

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
// Edge records go to the polygon's own slots (poly*emax ..) while the stream runs and
// are made visible by the header written at the end (n_edges == 0: nothing to draw).
//
// SCREEN = true starts at the X stage with already-projected tuples (the G_*_polygon
// entry points); polys[] then holds {first, length, colour, tex} descriptors.
template <int DIM, bool SCREEN>
__global__ void __launch_bounds__(SETUP_THREADS) setup_kernel(SetupArgs a)
{
    constexpr int NA = DIM - 2;
    constexpr int ESZ = edge_ints(NA);
    int poly = blockIdx.x * blockDim.x + threadIdx.x;
    if (poly >= a.np) return;

    PrimHdr hdr;
    hdr.yy = 0; hdr.xx = 0; hdr.colour = 0; hdr.meta = 0;

    const int4 pd = ((const int4 *)a.polys)[poly];
    int colour, tex, e, first;
    if (SCREEN) { first = pd.x; e = pd.y; colour = pd.z; tex = pd.w; }
    else        { colour = pd.x; tex = pd.y; e = pd.z; first = pd.w; }

    bool bad = false;        // polygon cannot be processed (limits); counted, not drawn
    bool unsupported = false;
    int kind = a.render_type;
    if (!SCREEN && kind == KIND_TEX_LIN && tex < 0) kind = KIND_SHADED;   // EngBase.cpp:470,518-522

    bool go = true;
    if (e < 2 || e + 1 > 16 || e > a.max_edges || first < 0 || (long long)first + e + 1 > (long long)a.npv) {
        bad = true; go = false;
    }
    if (kind == KIND_TEX_LIN && (tex < 0 || tex >= 64)) { unsupported = true; go = false; }

    // ---- pass 1 (object mode): extent of the e+1 tuples, C_volume_clipping (Clipp3D.cpp:46-58)
    bool zclip = false;
    if (!SCREEN && go) {
        int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
        for (int i = 0; i <= e; i++) {
            int idx = a.polyverts[(size_t)(first + i) * 6];
            if ((unsigned)idx >= (unsigned)a.nv) { bad = true; go = false; break; }
            int4 p = load_vertex(a, idx);
            mn[0] = min(mn[0], p.x); mx[0] = max(mx[0], p.x);
            mn[1] = min(mn[1], p.y); mx[1] = max(mx[1], p.y);
            mn[2] = min(mn[2], p.z); mx[2] = max(mx[2], p.z);
        }
        if (go) {
            if ((mx[2] < mn[0]) || (mx[2] < mn[1]) || (mx[2] < -mx[0]) || (mx[2] < -mx[1]) || (mx[2] <= 10))
                go = false;
            else
                zclip = mn[2] < 10;
        }
    }

    // ---- the stream
    int *erec = a.edges + (size_t)poly * a.emax * ESZ;
    int ne = 0;                       // edge records written
    int miny = INT_MAX, maxy = INT_MIN, minx = INT_MAX, maxx = INT_MIN;
    Tup<DIM> zprev, zfirst;           // Z stage: previous input tuple / first output tuple
    Tup<DIM> xprev, sprev, sfirst;    // X stage: previous input; S stage: previous input, first
    zprev = Tup<DIM>(); zfirst = zprev; xprev = zprev; sprev = zprev; sfirst = zprev;
    int zn = 0, xn = 0, sn = 0;       // tuples emitted by the Z stage / taken by X / taken by S
    int px[3] = {0, 0, 0}, py[3] = {0, 0, 0};
    int first_view_z = 0;             // clipped[2] of EngBase.cpp:492
    const int wx = a.W - 1, hy = a.H - 1, xc = a.W / 2, yc = a.H / 2;

    // object mode: iterations 0..e read the tuples, iteration e+1 is the closing event
    // of a z-clipped polygon.  screen mode: iterations 0..e read the tuples.
    const int iters = SCREEN ? e + 1 : e + 2;
    for (int it = 0; go && it < iters; it++) {
        Tup<DIM> zo0 = zprev, zo1 = zprev;      // up to two tuples leave the Z stage per step
        bool zclosing0 = false;
        int cz = 0;

        if (it <= e) {
            Tup<DIM> t;
            if (SCREEN) {
                const int *src = a.polyverts + (size_t)(first + it) * DIM;
#pragma unroll
                for (int j = 0; j < DIM; j++) t.v[j] = src[j];
                zo0 = t; zclosing0 = (it == e); cz = 1;
            } else {
                // CI_construct_tuples, EngBase.cpp:58-137
                const int *rec = a.polyverts + (size_t)(first + it) * 6;
                int4 p = load_vertex(a, rec[0]);
                t.v[0] = p.x; t.v[1] = p.y; t.v[2] = p.z;
                if (DIM >= 6) { t.v[3] = wshl(rec[1], 16); t.v[4] = wshl(rec[2], 16); t.v[5] = wshl(rec[3], 16); }
                if (DIM == 8) {
                    bool tx = kind == KIND_TEX_LIN;
                    t.v[DIM - 2] = tx ? wshl(rec[4], 16) : 0;
                    t.v[DIM - 1] = tx ? wshl(rec[5], 16) : 0;
                }
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
                    if (zn == 0) { zfirst = p; first_view_z = p.v[2]; }
                    zn++;
                    if (zn + 1 > 16) bad = true;
                }
                // T_perspective, TransBs.cpp:425-432
                int z = p.v[2];
                p.v[0] = wadd(persp_div(p.v[0], a.log_focus, z), xc);
                p.v[1] = wadd(persp_div(p.v[1], a.log_focus, z), yc);
                p.v[2] = (int)(-0x7ffff / z);
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
                if (clip_line<DIM, 0, true>(xprev, p, 0, wx, o1, o2, moved, bad)) {
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
                    if (clip_line<DIM, 1, true>(sprev, s, 0, hy, v1, v2, moved, bad)) {
                        int x1 = v1.v[0], y1 = v1.v[1], x2 = v2.v[0], y2 = v2.v[1];
                        miny = min(miny, min(y1, y2));
                        maxy = max(maxy, max(y1, y2));
                        int dx = x2 - x1, dy = y2 - y1;
                        dx = dx < 0 ? -dx : dx;
                        dy = dy < 0 ? -dy : dy;
                        int L = dx > dy ? dx : dy;
                        if (ne < a.emax) {
                            int *r = erec + ne * ESZ;
                            r[0] = x1 | (y1 << 16);
                            r[1] = x2 | (y2 << 16);
                            r[2] = wshl(v1.v[2], 16);
                            r[2 + NA] = L > 0 ? tdiv(wshl(wsub(v2.v[2], v1.v[2]), 16), L) : 0;
#pragma unroll
                            for (int q = 1; q < NA; q++) {
                                r[2 + q] = v1.v[2 + q];
                                r[2 + NA + q] = L > 0 ? tdiv(wsub(v2.v[2 + q], v1.v[2 + q]), L) : 0;
                            }
                        }
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
    if (draw && !SCREEN) {
        if (zn < 2) draw = false;
        else {
            // T_normal_z_negative, TransLn.cpp:218-223, on projected tuples 0,1,2
            long long c = (long long)wsub(px[1], px[0]) * (long long)wsub(py[2], py[0]) -
                          (long long)wsub(py[1], py[0]) * (long long)wsub(px[2], px[0]);
            if (!(c < 0)) draw = false;
        }
        if (draw && kind == KIND_TEX_LIN && (a.force_linear || first_view_z < 1024)) {
            // EngBase.cpp:491-493 routes this polygon to the perspective mapper
            unsupported = true; draw = false;
        }
    }
    if (ne > a.emax) { bad = true; draw = false; }

    bool reached_fill = draw;
    if (draw && !(miny < maxy && ne > 0)) draw = false;

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
        // one reference per tile of the bounding box (rows miny .. maxy-1)
        int tx0 = minx / TW, tx1 = maxx / TW, ty0 = miny / TH, ty1 = (maxy - 1) / TH;
        for (int ty = ty0; ty <= ty1; ty++)
            for (int tx = tx0; tx <= tx1; tx++) atomicAdd(&a.tile_count[ty * a.tiles_x + tx], 1u);
    }
    ((int4 *)a.hdr)[poly] = make_int4(hdr.yy, hdr.xx, hdr.colour, hdr.meta);

    // counters: one atomic per wave
    unsigned long long m;
    m = __ballot(reached_fill);
    if (m && (threadIdx.x & 63) == (__ffsll((long long)m) - 1)) atomicAdd(&a.status->polys_drawn, (unsigned long long)__popcll(m));
    m = __ballot(bad);
    if (m && (threadIdx.x & 63) == (__ffsll((long long)m) - 1)) atomicAdd(&a.status->overflow_polys, (unsigned long long)__popcll(m));
    m = __ballot(unsupported);
    if (m && (threadIdx.x & 63) == (__ffsll((long long)m) - 1)) atomicAdd(&a.status->unsupported_polys, (unsigned long long)__popcll(m));
}

// ------------------------------------------------------------------ binning

// Exclusive prefix sum of the per-tile reference counts (one workgroup; the tile count
// is at most a few hundred thousand).  Also zeroes the scatter cursors.
__global__ void __launch_bounds__(1024) scan_kernel(const unsigned *__restrict__ count, unsigned *__restrict__ offset,
                                                    unsigned *__restrict__ cursor, int n,
                                                    unsigned long long capacity, Status *status)
{
    __shared__ unsigned wsum[16];
    __shared__ unsigned carry_s;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        int i = base + t;
        unsigned v = i < n ? count[i] : 0u;
        unsigned s = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            unsigned o = __shfl_up(s, d);
            if (lane >= d) s += o;
        }
        if (lane == 63) wsum[w] = s;
        __syncthreads();
        unsigned pre = 0;
        for (int k = 0; k < w; k++) pre += wsum[k];
        unsigned carry = carry_s;
        if (i < n) { offset[i] = carry + pre + s - v; cursor[i] = 0; }
        __syncthreads();
        if (t == 1023) carry_s = carry + pre + s;
        __syncthreads();
    }
    if (t == 0) {
        offset[n] = carry_s;
        atomicAdd(&status->tile_refs, (unsigned long long)carry_s);
        if ((unsigned long long)carry_s > capacity)
            atomicAdd(&status->list_overflow, (unsigned long long)carry_s - capacity);
    }
}

// One polygon per lane: append its index to the list of every tile of its bounding
// box.  The order inside a list does not matter (the fill resolves by (depth, index)).
__global__ void __launch_bounds__(256) bin_kernel(const PrimHdr *__restrict__ hdr, int np, int tiles_x,
                                                  const unsigned *__restrict__ offset, unsigned *__restrict__ cursor,
                                                  unsigned *__restrict__ list, unsigned long long capacity)
{
    int poly = blockIdx.x * blockDim.x + threadIdx.x;
    if (poly >= np) return;
    int4 h = ((const int4 *)hdr)[poly];
    if ((h.w & 0xff) == 0) return;
    int miny = h.x & 0xffff, maxy = (h.x >> 16) & 0xffff, minx = h.y & 0xffff, maxx = (h.y >> 16) & 0xffff;
    int tx0 = minx / TW, tx1 = maxx / TW, ty0 = miny / TH, ty1 = (maxy - 1) / TH;
    for (int ty = ty0; ty <= ty1; ty++)
        for (int tx = tx0; tx <= tx1; tx++) {
            int tile = ty * tiles_x + tx;
            unsigned slot = offset[tile] + atomicAdd(&cursor[tile], 1u);
            if ((unsigned long long)slot < capacity) list[slot] = (unsigned)poly;
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
template <int NA>
__device__ __forceinline__ bool row_ends(const int *__restrict__ er, int ne, int y, int &beg, int &end,
                                         int (&S)[NA], int (&E)[NA])
{
    constexpr int ESZ = edge_ints(NA);
    int bx = INT_MAX, ex = INT_MIN, be = 0, bi = 0, ee = 0, ei = 0;
    for (int e = 0; e < ne; e++) {
        const int2 xy = *(const int2 *)(er + e * ESZ);
        int x1 = xy.x & 0xffff, y1 = (xy.x >> 16) & 0xffff, x2 = xy.y & 0xffff, y2 = (xy.y >> 16) & 0xffff;
        int dx = x2 - x1, dy = y2 - y1;
        int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
        int j = dy < 0 ? y1 - y : y - y1;
        if ((unsigned)j > (unsigned)ady) continue;
        int ia, ib, xa, xb;                 // point index and x of the row's min-x / max-x point
        if (adx > ady) {
            int L = adx, s = ady, ilo, ihi;
            if (s == 0) { ilo = 0; ihi = L; }
            else {
                ilo = j == 0 ? 0 : (2 * L * j - L + 2 * s - 1) / (2 * s);
                ihi = j == s ? L : (2 * L * (j + 1) - L + 2 * s - 1) / (2 * s) - 1;
            }
            if (dx >= 0) { ia = ilo; ib = ihi; xa = x1 + ilo; xb = x1 + ihi; }
            else         { ia = ihi; ib = ilo; xa = x1 - ihi; xb = x1 - ilo; }
        } else {
            int L = ady, s = adx;
            int h = L == 0 ? 0 : (2 * s * j + L) / (2 * L);
            ia = ib = j;
            xa = xb = dx < 0 ? x1 - h : x1 + h;
        }
        if (xa < bx) { bx = xa; be = e; bi = ia; }
        if (ex < xb) { ex = xb; ee = e; ei = ib; }
    }
    if (bx == INT_MAX) return false;
    const int *rb = er + be * ESZ + 2, *re = er + ee * ESZ + 2;
#pragma unroll
    for (int q = 0; q < NA; q++) {
        S[q] = wadd(rb[q], wmul(bi, rb[NA + q]));
        E[q] = wadd(re[q], wmul(ei, re[NA + q]));
    }
    beg = bx;
    end = ex;
    return true;
}

__device__ __forceinline__ unsigned long long depth_key(int z, unsigned id)
{
    return ((unsigned long long)((unsigned)z ^ 0x80000000u) << 32) | id;
}

// One workgroup per screen tile.  LDS holds the tile's packed (depth, draw index) keys
// and its colours.  The tile's polygon list is taken 256 polygons at a time; their rows
// inside the tile are numbered through a prefix sum and dealt one row per lane, 256 rows
// per round.  Round = phase A: every lane walks its span and lowers the keys with
// ds_min_u64; barrier; phase B: every lane walks its span again and, where the key is
// still its own, computes and stores the colour.  A later round or chunk that wins a
// pixel overwrites the colour in its own phase B, which the barrier orders after this
// one.  The frame is read from HBM once per tile (or not at all right after a clear)
// and written once.
template <int NA>
__global__ void __launch_bounds__(RASTER_THREADS) raster_kernel(RasterArgs a)
{
    constexpr int ESZ = edge_ints(NA);
    constexpr int KP = TW + 1;       // key row pitch in 8-byte words: lanes on consecutive rows hit different banks
    __shared__ unsigned long long keys[TH * KP];
    __shared__ int cols[TH * TW];
    __shared__ int c_poly[RASTER_THREADS];
    __shared__ int c_row0[RASTER_THREADS];
    __shared__ int c_off[RASTER_THREADS + 1];
    __shared__ int wsum[RASTER_THREADS / 64];

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int tile = blockIdx.x;
    const int tx0 = (tile % a.tiles_x) * TW, ty0 = (tile / a.tiles_x) * TH;
    const unsigned lbeg = a.tile_offset[tile];
    unsigned lcnt = a.tile_fill[tile];
    if ((unsigned long long)lbeg + lcnt > a.list_capacity)      // list overflow (reported by the scan): stay inside the list
        lcnt = a.list_capacity > lbeg ? (unsigned)(a.list_capacity - lbeg) : 0u;
    if (lcnt == 0 && !a.clear_pending) return;       // untouched tile of a frame that is already in HBM

    // ---- tile in
    for (int p = t; p < TW * TH; p += RASTER_THREADS) {
        int lx = p % TW, ly = p / TW, gx = tx0 + lx, gy = ty0 + ly;
        int z = INT_MAX, c = a.clear_colour;
        if (!a.clear_pending && gx < a.W && gy < a.H) {
            z = a.zbuf[(size_t)gy * a.W + gx];
            c = a.cbuf[(size_t)gy * a.W + gx];
        }
        keys[ly * KP + lx] = depth_key(z, 0u);
        cols[ly * TW + lx] = c;
    }
    __syncthreads();

    unsigned long long frags = 0, qrows = 0;

    for (unsigned cbase = 0; cbase < lcnt; cbase += RASTER_THREADS) {
        // ---- chunk: one polygon per lane, rows inside the tile, prefix sum
        int poly = -1, r0 = 0, cnt = 0;
        if (cbase + t < lcnt) {
            poly = (int)a.tile_list[lbeg + cbase + t];
            int yy = a.hdr[poly].yy;
            int miny = yy & 0xffff, maxy = (yy >> 16) & 0xffff;
            r0 = max(miny, ty0);
            int r1 = min(maxy, ty0 + TH);
            cnt = max(r1 - r0, 0);
        }
        int s = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int o = __shfl_up(s, d);
            if (lane >= d) s += o;
        }
        if (lane == 63) wsum[wv] = s;
        __syncthreads();
        int pre = 0;
        for (int k = 0; k < wv; k++) pre += wsum[k];
        c_poly[t] = poly;
        c_row0[t] = r0;
        c_off[t] = pre + s - cnt;
        if (t == RASTER_THREADS - 1) c_off[RASTER_THREADS] = pre + s;
        __syncthreads();
        const int total = c_off[RASTER_THREADS];

        for (int rbase = 0; rbase < total; rbase += RASTER_THREADS) {
            const int r = rbase + t;
            bool act = r < total;
            int y = 0, xa = 0, xb = -1, beg = 0, kind = 0, pcol = 0, ptex = 0;
            unsigned id = 0;
            int S[NA], I[NA];
#pragma unroll
            for (int q = 0; q < NA; q++) { S[q] = 0; I[q] = 0; }
            if (act) {
                // which polygon owns row r: last p with c_off[p] <= r
                int lo = 0, hi = RASTER_THREADS - 1;
                while (lo < hi) {
                    int mid = (lo + hi + 1) >> 1;
                    if (c_off[mid] <= r) lo = mid; else hi = mid - 1;
                }
                const int pp = c_poly[lo];
                y = c_row0[lo] + (r - c_off[lo]);
                const int4 h = ((const int4 *)a.hdr)[pp];
                const int ne = h.w & 0xff;
                kind = (h.w >> 8) & 0xff;
                ptex = (h.w >> 16) & 0xff;
                pcol = h.z;
                id = (unsigned)pp + 1u;
                int E[NA], end = 0;
                act = row_ends<NA>(a.edges + (size_t)pp * a.emax * ESZ, ne, y, beg, end, S, E);
                if (act) {
                    // span set-up, GraphicsPoly.cpp:335-353 (flat: :269-274)
                    int span = end - beg;
                    if (span == 0) span = 1;
#pragma unroll
                    for (int q = 0; q < NA; q++) I[q] = tdiv(wsub(E[q], S[q]), span);
                    if (NA == 6 && kind == KIND_TEX_LIN && ((I[1] | I[2] | I[3]) != 0)) qrows++;
                    xa = max(beg, tx0);
                    xb = min(end, tx0 + TW - 1);
                    if (xa > xb) act = false;
                    else frags += (unsigned long long)(xb - xa + 1);
                }
            }
            // ---- phase A: depth
            if (act) {
                unsigned long long *krow = keys + (y - ty0) * KP - tx0;
                int z = wadd(S[0], wmul(xa - beg, I[0]));
                for (int x = xa; x <= xb; x++) {
                    atomicMin(&krow[x], depth_key(z >> 16, id));
                    z = wadd(z, I[0]);
                }
            }
            __syncthreads();
            // ---- phase B: colour of the fragments that hold their pixel
            if (act) {
                const unsigned long long *krow = keys + (y - ty0) * KP - tx0;
                int *crow = cols + (y - ty0) * TW - tx0;
                int z = wadd(S[0], wmul(xa - beg, I[0]));
                for (int x = xa; x <= xb; x++) {
                    if (krow[x] == depth_key(z >> 16, id)) {
                        int c;
                        if (NA == 1 || kind == KIND_FLAT) c = pcol;
                        else {
                            const int k = x - beg;
                            int lr = wadd(S[1 % NA], wmul(k, I[1 % NA])) >> 16;
                            int lg = wadd(S[2 % NA], wmul(k, I[2 % NA])) >> 16;
                            int lb = wadd(S[3 % NA], wmul(k, I[3 % NA])) >> 16;
                            int base = pcol;
                            if (NA == 6 && kind == KIND_TEX_LIN) {
                                // GraphicsPoly.cpp:465,486: wrap, then fetch the texel
                                const int lg2 = a.tex_log[ptex];
                                const int mask = wsub(wshl(1, lg2 + 16), 1);
                                int tu = wadd(S[4 % NA], wmul(k, I[4 % NA])) & mask;
                                int tv = wadd(S[5 % NA], wmul(k, I[5 % NA])) & mask;
                                base = a.tex_ptr[ptex][((tv >> 16) << lg2) + (tu >> 16)];
                            }
                            c = cl_light(base, lr, lg, lb);
                        }
                        crow[x] = c;
                    }
                    z = wadd(z, I[0]);
                }
            }
            // the next round's phase A may run beside this phase B: it only lowers keys,
            // and a lane that loses its pixel here skips the store
        }
        __syncthreads();        // the chunk tables are rewritten next
    }
    __syncthreads();

    // ---- tile out
    for (int p = t; p < TW * TH; p += RASTER_THREADS) {
        int lx = p % TW, ly = p / TW, gx = tx0 + lx, gy = ty0 + ly;
        if (gx < a.W && gy < a.H) {
            unsigned long long k = keys[ly * KP + lx];
            a.zbuf[(size_t)gy * a.W + gx] = (int)((unsigned)(k >> 32) ^ 0x80000000u);
            a.cbuf[(size_t)gy * a.W + gx] = cols[ly * TW + lx];
        }
    }

    // counters: one atomic per wave
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        frags += __shfl_down(frags, d);
        qrows += __shfl_down(qrows, d);
    }
    if (lane == 0) {
        if (frags) atomicAdd(&a.status->fragments, frags);
        if (qrows) atomicAdd(&a.status->order_dependent_rows, qrows);
    }
}

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

void launch_setup(hipStream_t s, int dim, bool screen, const SetupArgs &a)
{
    if (a.np <= 0) return;
    dim3 g((a.np + SETUP_THREADS - 1) / SETUP_THREADS), b(SETUP_THREADS);
    if (screen) {
        if (dim == 3) hipLaunchKernelGGL((setup_kernel<3, true>), g, b, 0, s, a);
        else if (dim == 6) hipLaunchKernelGGL((setup_kernel<6, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((setup_kernel<8, true>), g, b, 0, s, a);
    } else {
        if (dim == 3) hipLaunchKernelGGL((setup_kernel<3, false>), g, b, 0, s, a);
        else if (dim == 6) hipLaunchKernelGGL((setup_kernel<6, false>), g, b, 0, s, a);
        else hipLaunchKernelGGL((setup_kernel<8, false>), g, b, 0, s, a);
    }
}

void launch_scan(hipStream_t s, const unsigned *count, unsigned *offset, unsigned *cursor, int ntiles,
                 unsigned long long capacity, Status *status)
{
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, count, offset, cursor, ntiles, capacity, status);
}

void launch_bin(hipStream_t s, const PrimHdr *hdr, int np, int tiles_x, const unsigned *offset,
                unsigned *cursor, unsigned *list, unsigned long long capacity, Status *)
{
    if (np <= 0) return;
    hipLaunchKernelGGL(bin_kernel, dim3((np + 255) / 256), dim3(256), 0, s, hdr, np, tiles_x, offset, cursor, list,
                       capacity);
}

void launch_raster(hipStream_t s, int na, const RasterArgs &a)
{
    dim3 g(a.tiles_x * a.tiles_y), b(RASTER_THREADS);
    if (na == 1) hipLaunchKernelGGL((raster_kernel<1>), g, b, 0, s, a);
    else if (na == 4) hipLaunchKernelGGL((raster_kernel<4>), g, b, 0, s, a);
    else hipLaunchKernelGGL((raster_kernel<6>), g, b, 0, s, a);
}

void launch_fill(hipStream_t s, int *cbuf, int *zbuf, size_t n, int colour)
{
    size_t n4 = n / 4;
    if (n4) {
        size_t blocks = (n4 + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (int4 *)cbuf, (int4 *)zbuf, n4, colour);
    }
    if (n4 * 4 < n) hipLaunchKernelGGL(fill_tail_kernel, dim3(1), dim3(4), 0, s, cbuf, zbuf, n4 * 4, n, colour);
}

}  // namespace lt
