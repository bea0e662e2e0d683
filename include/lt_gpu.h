/* lt_gpu.h -- C-ABI of the MI355X polygon rasterization library (libltgpu.so).
 *
 * Drop-in boundary for the polygon path of LightTrack / 3DGPL: the host-side engine
 * keeps the reference's API (Engine.h, Graphics.h, Trans.h, Clipper.h, Colour.h) and
 * forwards the data-parallel part to these entry points.  Plain pointers and sizes
 * only; every function returns 0 on success or a negative LT_ERR_* code, and
 * lt_last_error() gives the message (the reference itself has no error returns: it
 * calls HW_error and exits, LightTrack.cpp:56-69; the host mirror maps a non-zero
 * status to that behaviour).
 *
 * One context per GPU and per framebuffer; one host thread per context; all work of a
 * context is issued on one HIP stream, in call order.  Draw order is call order, and
 * inside one submit it is array order, as in M_render_polygon_object
 * (EngPoly.cpp:518-523): where two fragments have equal depth the earlier polygon
 * keeps the pixel (strict compare, GraphicsPoly.cpp:357).
 *
 * Flat scene layout (all int32, little endian):
 *   verts      [nv][4]   X Y Z and an unused 4th lane (T_LNG_VECTOR, Trans.h:27)
 *   polys      [np][4]   {colour 0x00RRGGBB, texture handle or -1, no_edges, first}
 *                        (m_colour, m_texture, m_no_edges, m_vertices of struct
 *                        M_polygon, Engine.h:209-230)
 *   polyaux    [np][4]   {m_red, m_green, m_blue, m_type | m_log_texture_space_size<<8};
 *                        needed for LT_FLAT; may be NULL otherwise
 *   polyverts  [*][6]    M_polygon.m_vertices records [Idx R G B Tx Ty]
 *                        (M_LNG_POLYGON_VERTEX, Engine.h:204-207); polygon i owns records
 *                        first .. first+no_edges (the closing vertex is stored)
 *   screen tuples [*][dim]  X Y Z' [R G B [Tx Ty]] as T_perspective leaves them
 *                        (dim 3, 6 or 8: G_LNG_FLAT/SHADED/TEXTURED, Graphics.h:54-58),
 *                        R G B Tx Ty in 16.16 fixed point
 *   framebuffer          colour int32[H][W] 0x00RRGGBB and depth int32[H][W], row stride W
 *                        (G_c_buffer / G_z_buffer, GraphicsBase.cpp:35-42)
 */
#ifndef LT_GPU_H
#define LT_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lt_ctx lt_ctx;

/* rendering types: M_WIRE / M_FLAT / M_SHADED / M_TEXTURED, Engine.h:80-83 */
#define LT_WIRE      1
#define LT_FLAT      2
#define LT_SHADED    4
#define LT_TEXTURED  8

#define LT_OK                 0
#define LT_ERR_ARG           -1   /* bad argument */
#define LT_ERR_HIP           -2   /* a HIP call failed */
#define LT_ERR_OVERFLOW      -3   /* a clipped polygon needs more than 16 tuples (G_MAX_POLYGON_VERTICES, Graphics.h:37) */
#define LT_ERR_CAPACITY      -4   /* tile list capacity exceeded: raise it with lt_set_list_capacity */
#define LT_ERR_UNSUPPORTED   -5   /* input needs a part of the reference this library does not cover (see message) */

#define LT_MAX_TEXTURES      64
#define LT_MAX_SCREEN     16384   /* width and height limit */

/* counters of the work done since the last lt_reset_counters */
typedef struct lt_counters {
    uint64_t fragments;        /* Z tests = pixels of all spans of all drawn polygons (the inner loop of GraphicsPoly.cpp:355-381) */
    uint64_t polys_in;         /* polygons submitted */
    uint64_t polys_drawn;      /* polygons that reached the fill (passed volume clip and back-face cull) */
    uint64_t tile_refs;        /* (polygon, screen tile) pairs binned */
    uint64_t overflow_polys;   /* polygons dropped for LT_ERR_OVERFLOW */
    uint64_t order_dependent_rows; /* textured rows whose light varies along the span (GraphicsPoly.cpp:485-493) */
    uint64_t unsupported_polys;
    uint64_t list_overflow;    /* tile references that did not fit */
    uint64_t zero_light_vectors; /* shading: (vertex, point light) pairs at zero distance, that light left out */
} lt_counters;

/* --- context ------------------------------------------------------------------ */

/* Replaces G_init_graphics (GraphicsBase.cpp:70-89): allocates the colour and depth
 * buffers in HBM for a width x height screen (a run-time value here; the reference
 * fixes it at compile time, LightTrack.h:44-45). */
int lt_ctx_create(int device, int width, int height, lt_ctx **out);
int lt_ctx_destroy(lt_ctx *ctx);
const char *lt_last_error(lt_ctx *ctx);          /* ctx may be NULL: last creation error */

/* The caller's hipStream_t (NULL = the default stream): device inputs of a submit may be
 * produced on it (each batch waits for what that stream holds at the time of the submit),
 * lt_transform and the host-array copies run on it.  Results are defined after lt_flush. */
int lt_set_stream(lt_ctx *ctx, void *hip_stream);
/* on (default): set-up and binning of a batch run on a stream of the context's own while
 * the fill of the batch before it runs on another, with two sets of work arrays; draw
 * order is kept.  off: every kernel is issued on the caller's stream, in order, with one
 * set in use.  No counterpart in the reference (which draws synchronously). */
int lt_set_pipelining(lt_ctx *ctx, int on);
/* Capacity, in (polygon, tile) references, of the per-submit tile lists.  0 = default
 * (64 per polygon, at least 1M). */
int lt_set_list_capacity(lt_ctx *ctx, uint64_t refs);
/* Width of the screen tiles of the fill for the batches submitted next: 64 or 128 pixels (tiles
 * are 32 rows high), or 0 (default) to let the library choose -- wide tiles once the polygons
 * drawn between two lt_flush calls averaged 256 depth tests or more.  The frame does not
 * depend on it.  No counterpart in the reference. */
int lt_set_tile_width(lt_ctx *ctx, int width);
/* Waves that share one 64-pixel tile of the fill: 4 (the rule), 8, or 0 = chosen by the library from
 * the tile lists of the batches before the last lt_flush (8 when polygons crowd into few tiles, e.g. an
 * object drawn small: the fill lasts as long as the longest list, and more waves shorten that).  No
 * counterpart in the reference; the frame and the counters do not depend on it. */
int lt_set_tile_waves(lt_ctx *ctx, int waves);
/* Host arrays and the way to the device.  The entry points that take host arrays (lt_submit_object,
 * lt_submit_objects, lt_submit_screen_polys*, lt_submit_lines, lt_shade_*, lt_plane_normals, lt_readback)
 * move arrays of 1 MB and more through a ring of pinned memory that `threads` host threads (the
 * calling one and threads - 1 helpers of the context; 1..16, default 4) fill or empty while the copy
 * engine works on the other half: 160 MB arrive in 3.5 ms instead of the 6.0 ms of HIP's own copy
 * from pageable memory, a 1080p frame comes back in 0.57 ms instead of 0.90.  The helpers leave the
 * arrays' cache lines in other cores' caches: a caller that writes the arrays anew every frame finds
 * those writes slower afterwards (measured: 160 MB rewritten in 25 ms instead of 14) and should keep
 * such arrays in pinned memory (below), or set 1.  Arrays in pinned memory (lt_host_alloc,
 * hipHostMalloc, hipHostRegister) go to the copy engine as they are: no host thread touches them.
 * The frame does not depend on any of this.  No counterpart in the reference. */
int lt_set_copy_threads(lt_ctx *ctx, int threads);
/* Pinned host memory for arrays that are handed to the entry points above frame after frame (160 MB
 * arrive in 2.9 ms, the rate of the link), usable from every GPU of the process; needs no context
 * (errors are reported through lt_last_error(NULL)).  lt_host_free(NULL) is allowed.  Free before the
 * process ends its use of HIP, not from a destructor that runs at exit.  No counterpart in the
 * reference (its arrays are the C heap's). */
int lt_host_alloc(size_t bytes, void **out);
int lt_host_free(void *p);

/* The reference's textured fills advance the light accumulators only where the depth
 * test passes (GraphicsPoly.cpp:485-493, 666-674), so with a light gradient along a span
 * a pixel's colour depends on the draw history of its row.  LT_ORDER_AUTO (default):
 * a textured batch in which some drawn polygon has differing vertex lights is filled by
 * the draw-order kernel (exact, slower), any other batch by the tile kernel.
 * LT_ORDER_ALWAYS: every textured batch takes the draw-order kernel.  LT_ORDER_NEVER:
 * the tile kernel always; affected pixels are counted and lt_flush reports
 * LT_ERR_UNSUPPORTED. */
#define LT_ORDER_AUTO    0
#define LT_ORDER_ALWAYS  1
#define LT_ORDER_NEVER   2
/* The draw-order kernel works on bands of rows.  For batches of 20 000 polygons or more each
 * band gets the list of its polygons in draw order first (a stable counting sort on the
 * device), and spans of up to 16 pixels are drawn 64 at a time; smaller batches, and screens
 * of more than 1024 bands (a band is as many rows as fit 150 KB of LDS at 8 bytes a pixel: 4
 * rows at 3840 pixels, 1 row from 9600 pixels on), let every band read every header.  Either flag,
 * or-ed into the mode, overrides the choice; the frame does not depend on it. */
#define LT_ORDER_LISTS_ALWAYS 4
#define LT_ORDER_LISTS_NEVER  8
int lt_set_textured_order(lt_ctx *ctx, int mode);

/* --- framebuffer --------------------------------------------------------------- */

/* G_clear (GraphicsBase.cpp:99-111) with the lit colour already computed
 * (CL_light(colour,r,g,b), Colour.cpp:93-102): colour buffer <- colour, depth <- INT_MAX.
 * The fill is deferred into the first rasterization that follows. */
int lt_clear(lt_ctx *ctx, int32_t lit_colour);
/* Blocks until all submitted work is done and reports any error it raised.  This is
 * the point where the reference's HW_blit would read the frame (LightTrack.cpp:284-298). */
int lt_flush(lt_ctx *ctx);
/* lt_flush, then copy the frame to host memory (either pointer may be NULL). */
int lt_readback(lt_ctx *ctx, int32_t *colour, int32_t *depth);
/* Device addresses of the frame, valid until lt_ctx_destroy; contents are current after lt_flush.
 * (The frame stays at these addresses from this call on: see lt_set_frame_overlap.) */
int lt_framebuffer(lt_ctx *ctx, int32_t **d_colour, int32_t **d_depth);
/* Two frames in flight.  G_clear (GraphicsBase.cpp:99-111) makes a frame independent of
 * everything drawn before it, so while the context draws into its own arrays each
 * lt_clear that follows a draw starts the new frame in the other of two array pairs, on a
 * fill stream of its own: its fills do not wait for the last fills of the frame before.
 * What lt_readback / lt_flush / counters show is unchanged (they wait for everything and
 * read the current frame); a frame that is cleared away without having been read is still
 * drawn in full.  Not used with a bound frame (lt_bind_framebuffer), with a band
 * (lt_set_band), without pipelining, or after lt_framebuffer has handed out the frame's
 * addresses.  Default: on if the environment variable GPU_MAX_HW_QUEUES (HIP's number of hardware
 * queues per process, read when HIP starts; 4 when unset) is 8 or more, else off -- the second
 * frame needs streams with hardware queues of their own to pay; 1 / 0 here override that.
 * When this library is loaded into a process that has not set the variable and in which HIP has
 * not started yet (no HIP call so far), it sets the variable to 8 itself; a value already set
 * stays, and a process whose HIP runtime is already running keeps the queues it has.
 * No counterpart in the reference. */
int lt_set_frame_overlap(lt_ctx *ctx, int on);
/* Draw into caller-owned device arrays int32[height][width] (both, or both NULL to go
 * back to the context's own frame) instead of the arrays lt_ctx_create allocated: the
 * counterpart of handing the reference other G_c_buffer / G_z_buffer pointers
 * (GraphicsBase.cpp:35-42).  Flushes first.  The arrays must stay valid while bound; their
 * contents are taken as the current frame (lt_clear still clears). */
int lt_bind_framebuffer(lt_ctx *ctx, int32_t *d_colour, int32_t *d_depth);
/* Sort-first split of one frame over several contexts: this context draws only rows
 * [y0, y1) of the screen.  Clipping and every interpolant still refer to the whole screen
 * (GI_scan's spans live in one row each, GraphicsPoly.cpp:124-200, so rows are
 * independent once the edges are set up); rows outside the band are left untouched or
 * cleared, never drawn.  The fragment counters of the bands of one frame add up to the
 * whole frame's.  Default: the whole screen.  No counterpart in the reference. */
int lt_set_band(lt_ctx *ctx, int y0, int y1);
/* C_init_clipping (Clipp2D.cpp:36-42) / the result of C_set_bounds (:64-77): the 2-D clip
 * window [minx, maxx] x [miny, maxy] of the batches submitted after this call -- the bounds
 * of C_polygon_x_clipping, of the y clip inside GI_scan and of G_line's two clips.  Default:
 * the whole screen, (0, 0, width-1, height-1) (Clipper.h:19-23).  The window must lie on the
 * screen (the reference would write outside its buffers otherwise) and minx < maxx,
 * miny < maxy (C_set_bounds reports other windows as "cannot clip" and its callers draw
 * nothing).  lt_clear still clears the whole frame, as G_clear does. */
int lt_set_clip_window(lt_ctx *ctx, int minx, int miny, int maxx, int maxy);
int lt_get_counters(lt_ctx *ctx, lt_counters *out);   /* implies lt_flush */
int lt_reset_counters(lt_ctx *ctx);

/* Stage timing with HIP events on the streams the kernels run on (no counterpart in the
 * reference, which has no timers).  on = 1: every stage of every submit is bracketed by
 * events; any other value is a mask of the stages to time (bit k = stage k, e.g. 16 = the
 * fill only): an event costs a few microseconds of stream time, so time only what is
 * needed inside a measured region.  With a mask, bits 8 and up hold a sampling period n
 * (on = mask | n << 8, n > 1): only every n-th submit is timed.  lt_get_stage_ms implies
 * lt_flush and returns, summed since the last call, the milliseconds of {transform, setup,
 * scan, bin, fill} and the number of submits that were timed. */
#define LT_STAGES 5
int lt_set_profiling(lt_ctx *ctx, int on);
int lt_get_stage_ms(lt_ctx *ctx, double ms[LT_STAGES], int *submits);

/* m_texture_bytes / m_log_texture_size of struct M_texture (Engine.h:148-152): copies a
 * 2^log_size square of int32 texels to HBM under `handle` (0 .. LT_MAX_TEXTURES-1). */
int lt_texture_upload(lt_ctx *ctx, int handle, const int32_t *texels, int log_size);

/* --- the path ------------------------------------------------------------------- */

/* T_cancatinated_rotation (TransBs.cpp:373-393) with the matrix as an argument:
 * m = {cx1 cy1 cz1 tx, cx2 cy2 cz2 ty, cx3 cy3 cz3 tz} (T_c* / T_t* of
 * T_cancatinate_self_world, TransBs.cpp:325-361).  d_from and d_to are DEVICE
 * pointers to int32[n][4]; the 4th lane of d_to is left untouched. */
int lt_transform(lt_ctx *ctx, const int32_t m[12], const int32_t *d_from, int32_t *d_to, int n);

/* struct M_light (Engine.h:116-130, _RGB_): type M_AMBIENT 1, M_POINT 2, M_DIRECT 4
 * (Engine.h:112-114); param = position (point) or direction (direct), 4th element unused. */
typedef struct lt_light {
    int32_t type;
    int32_t param[4];
    int32_t red, green, blue;
} lt_light;
#define LT_MAX_LIGHTS 16

/* M_shade_polygon_object (EngPoly.cpp:464-493) on device arrays: vertices and normals are
 * rotated by the self rotation s9 = {sx1 sy1 sz1, sx2 sy2 sz2, sx3 sy3 sz3} of
 * T_set_self_rotation (TransBs.cpp:144-183) and the vertices moved by (x, y, z); then
 * M_shade_polygon (EngBase.cpp:325-364) gives every vertex record of every polygon its
 * R,G,B (d_polyverts[*][1..3] are overwritten) through M_shade_vertex (EngBase.cpp:220-288)
 * and the polygon its flat illumination (d_polyaux[p][0..2] = m_red, m_green, m_blue).
 * d_normals int32[nn][4] are the object's unit normals scaled by 1<<7 (m_normals of
 * M_polygon_object), d_polynormals[npv] the normal index of each record (m_normals of
 * M_polygon).  Runs on the caller's stream (lt_set_stream), so a submit that follows sees
 * the result.  A point light at zero distance from a vertex stops the reference
 * (T_unit_vector, TransLn.cpp:185); here that light is left out for that vertex and
 * lt_flush reports LT_ERR_UNSUPPORTED.  Polygons with records or indices out of range are
 * left alone and reported as LT_ERR_OVERFLOW. */
int lt_shade_object_dev(lt_ctx *ctx, const int32_t s9[9], int32_t x, int32_t y, int32_t z,
                        const int32_t *d_verts, int nv, const int32_t *d_normals, int nn,
                        const int32_t *d_polys, int np, int32_t *d_polyverts,
                        const int32_t *d_polynormals, int npv, int32_t *d_polyaux,
                        const lt_light *lights, int nlights);
/* The same on host arrays, as the reference's caller holds them: polyverts and polyaux are
 * copied to the device, shaded and copied back before the call returns (M_shade_polygon_object
 * writes into the polygons' own records, EngBase.cpp:339-352). */
int lt_shade_object(lt_ctx *ctx, const int32_t s9[9], int32_t x, int32_t y, int32_t z,
                    const int32_t *verts, int nv, const int32_t *normals, int nn,
                    const int32_t *polys, int np, int32_t *polyverts,
                    const int32_t *polynormals, int npv, int32_t *polyaux,
                    const lt_light *lights, int nlights);

/* G_line (GraphicsBase.cpp:165-253) for n lines in draw order: lines[i] = {x1, y1, z1, x2, y2,
 * z2, lit colour, 0} with projected end points (what T_perspective returns) and the colour
 * CL_light(colour, red, green, blue) of the reference's arguments.  Each line is clipped to
 * the screen (C_line_x_clipping, C_line_y_clipping) and walked point by point; a point is
 * drawn where its depth is less than the depth buffer's, which lines do not write; where
 * lines overlap the later one stays.  The reference clips a line with dimension 2, so the
 * depth of a clipped end point is read from the third word of a static store of the clipper
 * (Clipp2D.cpp:99-104, 132, 150): 0 in a program that draws only wireframes, a leftover of
 * the polygon clipper's calls after filled polygons.  This library uses 0.  Lines are
 * drawn after everything submitted before them and before everything submitted after.
 * The eighth word selects the primitive: 0 a line; 1 G_dot (GraphicsBase.cpp:133-153) at
 * (x1, y1) with depth z1: drawn if on the screen and nearer than the depth buffer; 2 G_pixel
 * (:117-123) at (x1, y1): written without a depth test (off the screen it is dropped; the
 * reference would write outside its buffer).  Neither looks at the clip window, as in the
 * reference.  3 a line that the caller has already clipped: its end points, depths included,
 * are taken as they are (both must lie on the screen, else the line is dropped and lt_flush
 * reports LT_ERR_OVERFLOW) -- for a caller that reproduces the reference's stale depth of
 * clipped end points itself, as the host mirror does with HW_track_clipper_stores.
 * The device array must be 16-byte aligned. */
int lt_submit_lines_dev(lt_ctx *ctx, const int32_t *d_lines, int n);
int lt_submit_lines(lt_ctx *ctx, const int32_t *lines, int n);

/* The shading loop of M_shade_bicubic (EngBcub.cpp:314-325): M_shade_vertex
 * (EngBase.cpp:220-288) for n points that are already in world space -- the reference
 * evaluates them from the rotated and translated control points -- with their normals
 * rotated by s9 here (T_self_rotation, EngBcub.cpp:291).  d_points, d_normals, d_rgb are
 * int32[n][4]; d_rgb[i] = {R, G, B, 0}.  Errors as lt_shade_object_dev. */
int lt_shade_points_dev(lt_ctx *ctx, const int32_t s9[9], const int32_t *d_points, const int32_t *d_normals, int n,
                        const lt_light *lights, int nlights, int32_t *d_rgb);
/* The same on host arrays; rgb is filled before the call returns. */
int lt_shade_points(lt_ctx *ctx, const int32_t s9[9], const int32_t *points, const int32_t *normals, int n,
                    const lt_light *lights, int nlights, int32_t *rgb);

/* The plane normals of M_init_polygon_object (EngPoly.cpp:392-398): for every polygon,
 * T_normal_plane (TransLn.cpp:150-171) of the vertices of its first three records:
 * d_plane_normals[p] = {nx, ny, nz, 0}, unit length scaled by 1<<7.  These are the m_normals
 * of an object of M_PLANNAR polygons (one normal per polygon, index = polygon); the list
 * handling of M_init_polygon_object (shared indices, the averaged vertex normals of M_CURVED
 * polygons, EngPoly.cpp:399-458) is sequential host work on top of them (host mirror:
 * M_init_polygon_object).  Runs on the caller's stream.  A zero-length normal stops the
 * reference (TransLn.cpp:166); here it is stored as {0,0,0,0} and lt_flush reports
 * LT_ERR_UNSUPPORTED. */
int lt_plane_normals_dev(lt_ctx *ctx, const int32_t *d_verts, int nv, const int32_t *d_polys, int np,
                         const int32_t *d_polyverts, int npv, int32_t *d_plane_normals);
/* The same on host arrays; plane_normals int32[np][4] is filled before the call returns and
 * a zero-length normal is reported by this call. */
int lt_plane_normals(lt_ctx *ctx, const int32_t *verts, int nv, const int32_t *polys, int np,
                     const int32_t *polyverts, int npv, int32_t *plane_normals);

/* Rows of the screen that each polygon of an object can touch: d_rows (DEVICE, int32[np][2]) gets
 * {first row, last row}, or {1, 0} for a polygon that draws nothing.  For a sort-first split of one
 * frame over several GPUs (lt_set_band): each GPU asks this for its share of the polygons, the GPUs
 * exchange polygon indices by band, and each then submits only the polygons that can touch its band,
 * in their original order.  No counterpart in the reference.  Where every vertex lies beyond the
 * near plane the rows are those of the projected vertices, found as M_render_polygon finds them
 * (view-volume test Clipp3D.cpp:46-58, T_perspective TransBs.cpp:425-432, T_normal_z_negative
 * TransLn.cpp:218-223 at EngBase.cpp:433-470); a polygon that meets the near plane, or with data that
 * lt_submit_object would report, gets every row: never fewer rows than the polygon is drawn on.
 * Arguments as for lt_submit_object_dev; runs on the caller's stream. */
int lt_polygon_rows_dev(lt_ctx *ctx, const int32_t m[12],
                        const int32_t *d_verts, int nv,
                        const int32_t *d_polys, int np,
                        const int32_t *d_polyverts, int npv,
                        int log_focus, int max_edges, int32_t *d_rows);

/* M_render_polygon_object (EngPoly.cpp:499-525) for one object of np polygons, after
 * the host has built the matrix (T_set_self_rotation + T_cancatinate_self_world):
 * transform, then per polygon what M_render_polygon does (EngBase.cpp:379-552) --
 * tuples, view-volume test, near-plane clip, perspective, back-face cull, screen
 * clip, edge scan, span fill with depth test; with LT_WIRE (EngBase.cpp:393-431) the
 * edges of the near-clipped polygon are drawn as lines (see lt_submit_lines; polyaux gives
 * m_red, m_green, m_blue, zeros without it).  render_type is what M_init_rendering
 * set; log_focus what M_set_camera set; force_linear is M_force_linear_tmapping.
 * max_edges bounds no_edges over the batch (0 = 15).  m == NULL means the vertices are
 * already in view space (the M_render_polygon level, EngBase.cpp:379).  Pointers are
 * HOST memory; the call copies them to the GPU and returns after enqueueing. */
int lt_submit_object(lt_ctx *ctx, const int32_t m[12],
                     const int32_t *verts, int nv,
                     const int32_t *polys, const int32_t *polyaux, int np,
                     const int32_t *polyverts, int npv,
                     int render_type, int log_focus, int force_linear, int max_edges);
/* The same with every array already in DEVICE memory (nothing is copied). */
int lt_submit_object_dev(lt_ctx *ctx, const int32_t m[12],
                         const int32_t *d_verts, int nv,
                         const int32_t *d_polys, const int32_t *d_polyaux, int np,
                         const int32_t *d_polyverts, int npv,
                         int render_type, int log_focus, int force_linear, int max_edges);

/* A sequence of M_render_polygon_object calls (EngPoly.cpp:499-525) as ONE batch: the reference
 * takes objects of fewer than 1024 vertices and at most 256 polygons (Engine.h:432-433, guard at
 * EngPoly.cpp:507-508), so a world of a million triangles is thousands of calls, and a GPU batch
 * per call would cost each its own chain of kernels.  Here the objects of one frame that share
 * render type, log_focus and force_linear are laid end to end in the same arrays as for
 * lt_submit_object and a table names, per object, the matrix that the host built for it
 * (T_set_self_rotation + T_cancatinate_self_world) and its range of the vertex array:
 *   objects int32[nobj][16]  lt_object: m[12], vert_base, vert_count, two reserved words (0)
 *   polyobj int32[np]        the object of every polygon; ascending, so that draw order (which
 *                            decides between equal depths, GraphicsPoly.cpp:357) is the order of
 *                            the calls, then the order inside each object
 *   polyverts                vertex indices count from the object's vert_base and must lie
 *                            below its vert_count; polys[p].first indexes polyverts as a whole
 * A polygon whose polyobj is outside the table, or whose object's range leaves the vertex array,
 * is not drawn and is reported like a vertex index out of range (LT_ERR_OVERFLOW at lt_flush).
 * render_type: LT_FLAT, LT_SHADED or LT_TEXTURED (wireframe objects go through
 * lt_submit_object one by one).  The frame is the same, pixel for pixel, as after one
 * lt_submit_object per object in table order.  Pointers are HOST memory. */
typedef struct lt_object {
    int32_t m[12];
    int32_t vert_base, vert_count;
    int32_t reserved[2];
} lt_object;
int lt_submit_objects(lt_ctx *ctx, const lt_object *objects, int nobj, const int32_t *polyobj,
                      const int32_t *verts, int nv,
                      const int32_t *polys, const int32_t *polyaux, int np,
                      const int32_t *polyverts, int npv,
                      int render_type, int log_focus, int force_linear, int max_edges);
/* The same with every array, the table included, already in DEVICE memory (nothing is copied). */
int lt_submit_objects_dev(lt_ctx *ctx, const lt_object *d_objects, int nobj, const int32_t *d_polyobj,
                          const int32_t *d_verts, int nv,
                          const int32_t *d_polys, const int32_t *d_polyaux, int np,
                          const int32_t *d_polyverts, int npv,
                          int render_type, int log_focus, int force_linear, int max_edges);

/* lt_submit_object_dev for callers that must not allocate device memory while they draw: tight
 * serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference.  (The call is
 * not itself capturable into a HIP graph: it issues on the library's own streams and queries the
 * caller's.)  Same arguments, same result in the frame and the counters (LT_WIRE excepted: not
 * supported here), but the batch's work arrays (polygon headers, edge records, tile lists,
 * set-up queues) are pieces of one static device array of LT_NOALLOC_SCRATCH_BYTES that each
 * GPU holds from the first lt_ctx_create on it until the process ends; the call makes no
 * hipMalloc / hipFree and uses no work set of the context.  The tile lists get the same room as
 * in lt_submit_object_dev (2^20 entries at least) unless only a smaller list lets the batch fit.
 * A batch that needs more than the array returns LT_ERR_CAPACITY with the sizes in the message and
 * draws nothing: about 55 000 Gouraud triangles per call at max_edges = 3 (roughly 1000 bytes
 * each with the set-up queues), fewer with more edges, a larger lt_set_list_capacity or a textured
 * batch.  All kernels run on the stream of the current frame's fills, in submit order with every
 * other submit of the context; stage timing (lt_set_profiling) skips these batches.  Every call
 * on a GPU shares that GPU's array: the library orders them one after the other, other contexts
 * and other host threads on the same GPU included (the enqueueing of a call is done under a lock). */
#define LT_NOALLOC_SCRATCH_BYTES (64u << 20)
int lt_submit_object_dev_noalloc(lt_ctx *ctx, const int32_t m[12],
                                 const int32_t *d_verts, int nv,
                                 const int32_t *d_polys, const int32_t *d_polyaux, int np,
                                 const int32_t *d_polyverts, int npv,
                                 int render_type, int log_focus, int force_linear, int max_edges);

/* G_flat_polygon / G_shaded_polygon / G_lin_textured_polygon (GraphicsPoly.cpp:230-502)
 * for n already-projected closed polygons, in array order.  kind = LT_FLAT (dim 3),
 * LT_SHADED (dim 6) or LT_TEXTURED (dim 8).  desc[i] = {first tuple, length (edges),
 * colour, texture handle}: for LT_FLAT colour is the lit colour CL_light(colour,r,g,b),
 * for LT_SHADED the base colour.  Polygon i reads tuples first .. first+length.
 * HOST pointers. */
int lt_submit_screen_polys(lt_ctx *ctx, int kind, const int32_t *tuples, int ntuples,
                           const int32_t *desc, int n, int max_edges);

/* G_prp_textured_polygon (Graphics.h:108-113, GraphicsPoly.cpp:520-684) for n already-projected
 * closed polygons in array order: tuples of dim 8 (X Y Z' R G B Tx Ty; the edge scan
 * interpolates Z' R G B only, GraphicsPoly.cpp:559), desc[i] = {first tuple, length (edges),
 * unused, texture handle}, and per polygon the function's other arguments
 *   prp[i][12] = {O[0], O[1], O[2], O[6], O[7], u[0], u[1], u[2], v[0], v[1], v[2],
 *                 log_texture_scale}
 * -- O is the view-space tuple the texture origin maps to (X Y Z and its texture coordinates,
 * 16.16), u and v the texture orientation vectors of MI_tmapping_vectors (EngBase.cpp:145-179).
 * log_focus is M_camera_log_focus, which the reference's function reads as a global
 * (GraphicsPoly.cpp:614-633).  Texture coordinates are exact every 32 pixels of a span
 * (G_LINEAR, GraphicsPoly.cpp:40) and linear between; where the divisor zz is 0 the reference
 * uses 0 for both.  HOST pointers. */
int lt_submit_screen_polys_prp(lt_ctx *ctx, const int32_t *tuples, int ntuples, const int32_t *desc,
                               const int32_t *prp, int n, int log_focus, int max_edges);

#ifdef __cplusplus
}
#endif
#endif
