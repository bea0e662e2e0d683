// Device-side data layout and launch wrappers shared by the kernels (lt_kernels.hip)
// and the C-ABI (lt_api.cpp side of the same library).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

namespace lt {

// Screen tile owned by one workgroup of the fill kernel.  Its depth keys and colours
// live in LDS for the whole kernel.
constexpr int TH = 32;            // tile height (rows)
// Tile width: 64 pixels (256 threads a workgroup) or 128 (512 threads).  A span is walked once
// per tile it crosses, so wide tiles pay off when polygons are wide; narrow tiles leave the
// set-up kernels of the next batch more room beside the fill.  Chosen per batch.
constexpr int TW_NARROW = 64, TW_WIDE = 128;
constexpr int SETUP_THREADS = 128;

// Per-polygon header written by the setup kernels at the front of the polygon's record (16 bytes):
//   yy    = miny | maxy<<16   rows [miny, maxy) are filled (GraphicsPoly.cpp:329)
//   xx    = minx | maxx<<16   column range of the clipped polygon (for binning)
//   colour: LT_FLAT: lit colour; LT_SHADED: base colour; LT_TEXTURED: base colour
//   meta  = n_edges | kind<<8 | tex<<16 ; n_edges == 0 marks "nothing to draw"
// The binning kernels (tiles, bands) read their own dense array of the first two words only,
// 8 bytes a polygon: a polygon that is drawn has at least one row, so yy == 0 marks "nothing to draw" there.
struct PrimHdr { int yy, xx; };

// Edge record after the y clip of GI_scan (GraphicsPoly.cpp:136), in 16-byte pieces:
//   {x1|y1<<16, x2|y2<<16, Zstart, Zinc, start[1..NA-1], inc[1..NA-1], padding}
// Zstart = Z1<<16, Zinc = ((Z2-Z1)<<16)/L; the other values as in GraphicsPoly.cpp:171-176.
// NA = 1 (flat: 4 ints), 4 (shaded: 12 ints), 6 (textured: 16 ints).  The depth pass
// reads only the first piece.  The pool holds, per output slot, one record of
// 4 + 3*(pieces-1) 16-byte units: {copy of the header, first piece of edges 0, 1, 2, the
// other pieces of edge 0, of edge 1, of edge 2}, so a triangle's depth data is one 64-byte
// run and the whole triangle one run of 64 / 160 / 208 bytes.  Edges 3 and up (clipped
// polygons) follow the records as planes [edge][piece][slot].  No allocation is needed.
__host__ __device__ constexpr int edge_ints(int na) { return na == 1 ? 4 : (na == 4 ? 12 : 16); }

// Constants of G_prp_textured_polygon for one polygon (GraphicsPoly.cpp:563-578):
// {ui, uj, uc, vi, vj, vc, zi, zj, zc, u0, v0, log_texture_scale}
constexpr int PRP_INTS = 12;

// Device status block, read back at lt_flush.  There are STATUS_SLOTS copies and a
// workgroup adds into slot blockIdx % STATUS_SLOTS: thousands of waves adding to one
// word would queue behind each other (one chip-wide atomic per ~12 ns on one address);
// the host sums the slots.
constexpr int BIN_MAX_TILES = 8192;        // LDS histogram limit of the binning kernels; larger screens use direct atomics
constexpr int STATUS_SLOTS = 256;
constexpr int MAX_DEVICES = 64;            // device indices the library keeps per-device state for
constexpr int QUEUES = 64;                 // sub-queues between the fast and the general setup kernel
constexpr int QUEUE_COUNT_STRIDE = 32;     // words between their counters (one 128-byte line each)
// Workgroups of the fast set-up kernel.  Two waves (12 KB of LDS) and not four: while a fill is
// running its workgroups hold most of a CU's LDS, and a small set-up workgroup of the batch
// behind it still finds room beside them (C3 at 4K: 1.170 -> 1.148 ms per frame).
constexpr int SETUP_FAST_THREADS = 128;
// LDS that a fill launch asks for on top of its own when set-up chains of other batches run
// beside it: three fill workgroups a CU instead of four, and what they leave goes to set-up
// workgroups (C2-small with two frames in flight: 0.1306 -> 0.1271 ms per frame; the fill
// with the chip to itself is 13 % slower that way, so only launches that will be overlapped ask).
constexpr unsigned FILL_OVERLAP_LDS = 1024;
inline unsigned setup_queue_cap(int np)
{
    unsigned blocks = (unsigned)((np + SETUP_FAST_THREADS - 1) / SETUP_FAST_THREADS);
    return ((blocks + QUEUES - 1) / QUEUES) * SETUP_FAST_THREADS;
}
struct Status {
    unsigned long long fragments;
    unsigned long long polys_drawn;
    unsigned long long tile_refs;
    unsigned long long overflow_polys;
    unsigned long long order_dependent_rows;
    unsigned long long unsupported_polys;
    unsigned long long list_overflow;
    unsigned long long zero_light_vectors;   // shading: (vertex, point light) pairs at zero distance, skipped
    unsigned long long longest_lists;        // sum over the batches of each batch's longest tile list
};

// One object of a table of objects drawn by one batch (lt_submit_objects): its concatenated
// transform (T_cancatinate_self_world, TransBs.cpp:325-361) and its range of the shared vertex
// array; the vertex indices of its polygons' records count from vert_base.  Same layout as
// lt_object in lt_gpu.h.
struct ObjectDesc { int m[12]; int vert_base, vert_count, reserved0, reserved1; };

struct SetupArgs {
    int m[12];               // concatenated transform, applied while vertices are read
    int use_matrix;          // 0: verts are already in view space
    const ObjectDesc *objects;   // non-null: a table of objects; polyobj[p] names polygon p's object
    const int *polyobj;
    int nobj;
    const int *verts;        // int32[nv][4]                  (object mode)
    const int *polys;        // int32[np][4]                  (object mode) / desc (screen mode)
    const int *polyaux;      // int32[np][4] or null
    const int *polyverts;    // int32[*][6]                   (object mode) / tuples (screen mode)
    int np, nv, npv;
    int render_type, log_focus, force_linear, max_edges;
    int W, H, tiles_x, tiles_y;
    int emax;                // edge slots per polygon
    PrimHdr *hdr;
    int *edges;
    uint2 *queue;            // {polygon, output slot} left to the general kernel by the fast one (null: it takes all, slot = polygon)
    unsigned *queue_count;   // QUEUES counters, QUEUE_COUNT_STRIDE words apart
    unsigned queue_cap;      // entries per sub-queue
    unsigned queue_index_limit;  // queue entries name polygons below this (np, or the screen queue's size)
    int count_drawn;         // 0: polys_drawn has been counted by the kernel that queued the polygons
    // screen queue (object batches): projected triangles that meet a screen border
    int *sq_desc;            // int4 per entry {first tuple, 3 | kind<<8, colour, tex}
    int *sq_tuples;          // 4 tuples of DIM ints per entry
    uint2 *sq_queue;         // {entry, output slot}
    unsigned *sq_count;      // QUEUES counters, QUEUE_COUNT_STRIDE words apart
    int *order_flag;         // device word set to order_gen when a textured polygon's vertex lights differ
    int order_gen;           // this batch's number (never 0): a stale flag of an earlier batch does not count
    int detect_order;
    int *prp;                // int32[np][PRP_INTS]: perspective-mapper constants (textured object batches) or null
    int prp_given;           // screen batches: the caller supplied the constants of every polygon (indexed by polygon)
    int band_y0, band_y1;    // rows [band_y0, band_y1) are drawn by this context (whole screen: 0, H)
    int tile_w_log;          // log2 of the tile width of this batch (6 or 7)
    int cx0, cy0, cx1, cy1;  // 2-D clip window C_x_clipping_min .. C_y_clipping_max (whole screen: 0, 0, W-1, H-1)
    unsigned *tile_count;    // non-null: set-up counts the tile references itself (no count kernel)
    Status *status;
};

struct RasterArgs {
    const PrimHdr *hdr;
    const int *edges;
    const unsigned *tile_offset;   // exclusive prefix of tile_count, [ntiles+1]
    const unsigned *tile_fill;     // entries actually written per tile
    const unsigned *tile_list;
    unsigned long long list_capacity;
    int np;
    int emax;
    int W, H, tiles_x, tiles_y;
    int *cbuf, *zbuf;
    int clear_pending, clear_colour;
    const int *prp;
    int log_focus;
    const int *order_flag;   // see ordered_kernel: raised for this batch iff *order_flag == order_gen
    int order_gen, force_order;
    const int *const *tex_ptr;     // device table of texture base pointers
    const int *tex_log;            // device table of log sizes
    int band_y0, band_y1;          // rows drawn by this context
    int start_rank_shift;          // set by launch_raster: log2 of the device's CU count (0: no priority rotation)
    Status *status;
};

// Per-band polygon lists of the draw-order fill (textured batches whose light varies along
// spans): band b holds rows [b*R, b*R + R); its list is list[base[b] .. base[b] + total[b]).
constexpr int MAX_BANDS = 1024;            // LDS bit matrix of the band binning: MAX_BANDS x 33 words
struct BandArgs {
    const PrimHdr *hdr;
    int np;
    int R, nb;                     // rows per band, number of bands
    unsigned *cnt;                 // [chunks of 1024 slots][nb]: entries of the chunk, then their exclusive prefix along the chunks
    unsigned *base, *total;        // [nb]
    unsigned *list;
    unsigned long long capacity;   // entries
    unsigned *alloc;               // end of the lists handed out so far; zero before the batch, zeroed again by the fill
    const int *order_flag;
    int order_gen, force_order;
    Status *status;
};
bool band_lists_possible(int W, int H, int *R, int *nb);
hipError_t launch_band_lists(hipStream_t s, int np, int device, const BandArgs &bands);

// Shading pre-pass (M_shade_polygon_object, EngPoly.cpp:464-493).
constexpr int MAX_LIGHTS = 16;
struct ShadeArgs {
    int s[9];                // self rotation sx1 sy1 sz1 sx2 .. sz3
    int x, y, z;             // object position
    const int *verts;  int nv;       // int32[nv][4]
    const int *normals; int nn;      // int32[nn][4]
    const int *polys;  int np;       // int32[np][4] = {colour, tex, n_edges, first}
    int *polyverts;                  // int32[npv][6]: R,G,B written
    const int *polynormals; int npv; // normal index per record
    int *polyaux;                    // int32[np][4]: flat r,g,b written
    int nlights;
    int lights[MAX_LIGHTS][8];       // {type, px,py,pz,pw, red,green,blue}
    Status *status;
};
void launch_wire(hipStream_t s, const SetupArgs &a, const int *zbuf, unsigned *winner, int *colours, int *cbuf);
struct ClipWindow { int x0, y0, x1, y1; };
void launch_lines(hipStream_t s, const int *lines, int n, int W, int H, int by0, int by1, ClipWindow cw, const int *zbuf,
                  unsigned *winner, int *cbuf, Status *status);
void launch_plane_normals(hipStream_t s, const int *verts, int nv, const int *polys, int np, const int *polyverts,
                          int npv, int *out, Status *status);
void launch_shade(hipStream_t s, const ShadeArgs &a);
void launch_shade_points(hipStream_t s, const ShadeArgs &a, const int *points, const int *normals, int n, int *rgb);

void launch_transform(hipStream_t s, const int *m12_dev, const int *from, int *to, int n);
void launch_polygon_rows(hipStream_t s, const SetupArgs &a, int *rows);
void launch_setup(hipStream_t s, int dim, bool screen, const SetupArgs &a);
// tile lists: count -> exclusive scan -> scatter (count must be zero before the call)
void launch_binning(hipStream_t s, const PrimHdr *hdr, int np, int twl, int tiles_x, int ntiles, unsigned *count,
                    unsigned *offset, unsigned *cursor, unsigned *list, unsigned long long capacity,
                    unsigned *queue_count, Status *status, hipEvent_t after_scan, bool counted, hipEvent_t done,
                    const int *order_flag = nullptr, int order_gen = 0, int order_force = 0);
void launch_raster(hipStream_t s, int na, int tile_w, const RasterArgs &a, hipEvent_t start, hipEvent_t stop, unsigned extra_lds = 0, int waves = 4);
hipError_t launch_ordered(hipStream_t s, const RasterArgs &a, int device, const BandArgs *bands);   // textured batches only; `device` is current
void launch_fill(hipStream_t s, int *cbuf, int *zbuf, size_t n, int colour);

}  // namespace lt
