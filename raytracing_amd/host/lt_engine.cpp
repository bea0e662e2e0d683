// Host mirror of the reference's polygon-path API over the C-ABI (see lt_engine.h).
// Matrix set-up stays on the host, as in the reference (it is a few dozen integer
// multiplies per object); everything per vertex, per polygon and per pixel goes to the GPU.
#include "lt_engine.h"
#include "../../include/lt_gpu.h"

#include <limits.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <emmintrin.h>
#include <map>
#include <utility>
#include <vector>

// ------------------------------------------------------------------ state

HW_pixel *G_c_buffer = nullptr;
int *G_z_buffer = nullptr;
long G_c_buffer_size = 0;
int M_force_linear_tmapping = 0;                     // EngBase.cpp:36
int M_camera_x, M_camera_y, M_camera_z, M_camera_log_focus;

namespace {

int g_w = 600, g_h = 600;                            // the reference's compile-time size, LightTrack.h:44-45
int g_device = 0;
lt_ctx *g_ctx = nullptr;
int g_rendering_type = 0;
bool g_host_frame_current = false;
bool g_track_stores = false;                         // HW_track_clipper_stores

int T_sin_tab[256], T_cos_tab[256];
int Ts[9], Tw[9], Tc[12];                            // self / world rotation, concatenated transform
int Cb[4] = {0, 0, 0, 0};                            // clip window

// queued draws: consecutive calls of one kind form one batch
enum QKind { Q_NONE, Q_SCREEN, Q_VIEW, Q_LINES, Q_OBJECTS };
QKind q_kind = Q_NONE;
int q_type = 0;                                      // LT_FLAT / LT_SHADED / LT_TEXTURED
int q_focus = 0, q_force = 0;
std::vector<int32_t> q_tuples, q_desc;               // Q_SCREEN
std::vector<int32_t> q_prp;                          // Q_SCREEN with the perspective mapper: 12 words a polygon
constexpr int LT_TEXTURED_PRP = 9;                   // queue type of G_prp_textured_polygon (no C-ABI render type)
// (vectors whose resize() leaves new elements uninitialised: the queues of a frame of many objects grow by
// hundreds of megabytes that are written straight afterwards)
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    template <class U, class... A> void construct(U *p, A &&...a)
    {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U;
        else ::new ((void *)p) U(std::forward<A>(a)...);
    }
};
using IVec = std::vector<int32_t, NoInitAlloc<int32_t>>;
// The queues of view-space polygons and of whole objects are written anew every frame and handed to the
// library as host arrays.  Their larger blocks are pinned memory (lt_host_alloc): the copy engine reads
// pinned arrays as they are, at the rate of the link, where arrays of the C heap are first copied by host
// threads whose reads leave the lines in other cores' caches and make the next frame's writes slow.  Blocks
// that the library cannot pin (no GPU yet, or more than 1 GB at once) come from the C heap.  The queues live
// until the process ends and are never destroyed: at exit HIP may be gone before a destructor here runs.
std::map<void *, bool> &pinned_blocks()
{
    static std::map<void *, bool> *m = new std::map<void *, bool>;
    return *m;
}
template <class T>
struct QueueAlloc : NoInitAlloc<T> {
    template <class U> struct rebind { using other = QueueAlloc<U>; };
    T *allocate(size_t n)
    {
        const size_t bytes = n * sizeof(T);
        void *p = nullptr;
        if (bytes >= ((size_t)64 << 10) && bytes <= ((size_t)1 << 30) && lt_host_alloc(bytes, &p) == LT_OK) {
            pinned_blocks()[p] = true;
            return (T *)p;
        }
        p = malloc(bytes);
        if (!p) throw std::bad_alloc();
        return (T *)p;
    }
    void deallocate(T *p, size_t)
    {
        auto it = pinned_blocks().find(p);
        if (it == pinned_blocks().end()) { free(p); return; }
        pinned_blocks().erase(it);
        (void)lt_host_free(p);
    }
};
using QVec = std::vector<int32_t, QueueAlloc<int32_t>>;
QVec &q_verts = *new QVec, &q_polys = *new QVec, &q_aux = *new QVec, &q_pv = *new QVec;  // Q_VIEW (vertices already in view space)
std::vector<int32_t> q_lines;                        // Q_LINES: {x1 y1 z1 x2 y2 z2 lit-colour 0}
// Q_OBJECTS: consecutive M_render_polygon_object calls of one render type, laid end to end in
// q_verts / q_polys / q_aux / q_pv, with one lt_object per call (its matrix and vertex range)
// and each polygon's object: one GPU batch for all of them (lt_submit_objects)
std::vector<lt_object> q_objects;
QVec &q_polyobj = *new QVec;
int q_max_edges = 0;
// Textures are the caller's memory, read by the reference at draw time (GraphicsPoly.cpp:486).
// Here a texel array is uploaded once and known by its address; the first use of an address
// after a batch has been submitted checks a digest of the texels, so that an array whose
// contents changed -- or a new array that the allocator placed at an old address -- is uploaded
// again under the same handle.
struct TexEntry { int handle; int log_size; uint64_t digest; unsigned epoch; };
std::map<const void *, TexEntry> g_tex_handles;      // texel pointer -> handle
int g_next_tex = 0;
unsigned g_tex_epoch = 1;                            // advanced whenever a batch has been submitted

inline int wmul(int a, int b) { return (int)((unsigned)a * (unsigned)b); }
inline int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
inline int fx(int a, int b) { return wmul(a, b) >> T_P; }

void chk(int rc)
{
    if (rc != LT_OK) {
        char buf[600];
        snprintf(buf, sizeof buf, "(GPU) %s", lt_last_error(g_ctx));
        HW_error(buf);
    }
}

void need_ctx()
{
    if (!g_ctx) HW_error((char *)"(Graphics) G_init_graphics has not been called.\n");
}

uint64_t texel_digest(const HW_pixel *t, int log_size)
{
    // four interleaved multiply-xor chains over the words (one chain would wait for its own multiply)
    const size_t n = (size_t)1 << (2 * log_size);
    uint64_t h[4] = {0xcbf29ce484222325ull, 0x9e3779b97f4a7c15ull, 0xbf58476d1ce4e5b9ull, 0x94d049bb133111ebull};
    size_t i = 0;
    for (; i + 4 <= n; i += 4)
        for (int k = 0; k < 4; k++) h[k] = (h[k] ^ (uint32_t)t[i + k]) * 0x100000001b3ull;
    for (; i < n; i++) h[0] = (h[0] ^ (uint32_t)t[i]) * 0x100000001b3ull;
    return h[0] ^ (h[1] << 1) ^ (h[2] << 2) ^ (h[3] << 3) ^ (uint64_t)log_size;
}

void run_queue();

// `may_flush`: nothing of the draw that asks has been queued yet, so the open batch may be
// submitted to free a handle.
int texture_handle(HW_pixel *texels, int log_size, bool may_flush = false)
{
    auto it = g_tex_handles.find(texels);
    if (it != g_tex_handles.end() && it->second.epoch == g_tex_epoch && it->second.log_size == log_size) return it->second.handle;
    const uint64_t d = texel_digest(texels, log_size);
    if (it != g_tex_handles.end()) {
        TexEntry &e = it->second;
        if (e.digest != d || e.log_size != log_size) {
            chk(lt_texture_upload(g_ctx, e.handle, texels, log_size));     // waits for the batches that read the old texels
            e.digest = d; e.log_size = log_size;
        }
        e.epoch = g_tex_epoch;
        return e.handle;
    }
    int h;
    if (g_next_tex < LT_MAX_TEXTURES) {
        h = g_next_tex++;
    } else {
        // Every handle is taken (the reference knows a texture by its pointer and has no such limit):
        // the address that has gone unused for longest gives up its handle and is forgotten -- a
        // later draw with it uploads it again.  The textures of the batch being built keep theirs.
        auto oldest = [&]() {
            auto v = g_tex_handles.end();
            for (auto i = g_tex_handles.begin(); i != g_tex_handles.end(); ++i)
                if (i->second.epoch != g_tex_epoch && (v == g_tex_handles.end() || i->second.epoch < v->second.epoch)) v = i;
            return v;
        };
        auto victim = oldest();
        if (victim == g_tex_handles.end()) {
            if (!may_flush) HW_error((char *)"(GPU) more than 64 textures in one batch.\n");
            run_queue();
            victim = oldest();
        }
        h = victim->second.handle;
        g_tex_handles.erase(victim);
    }
    chk(lt_texture_upload(g_ctx, h, texels, log_size));       // waits for the batches that read the handle's old texels
    g_tex_handles[texels] = TexEntry{h, log_size, d, g_tex_epoch};
    return h;
}

void run_queue()
{
    bool submitted = true;
    _mm_sfence();                                    // the queues' streaming stores, before anyone else reads the arrays
    if (q_kind == Q_SCREEN && !q_desc.empty() && q_type == LT_TEXTURED_PRP) {
        chk(lt_submit_screen_polys_prp(g_ctx, q_tuples.data(), (int)(q_tuples.size() / 8), q_desc.data(), q_prp.data(),
                                       (int)(q_desc.size() / 4), q_focus, q_max_edges));
    } else if (q_kind == Q_SCREEN && !q_desc.empty()) {
        int dim = q_type == LT_FLAT ? 3 : (q_type == LT_SHADED ? 6 : 8);
        chk(lt_submit_screen_polys(g_ctx, q_type, q_tuples.data(), (int)(q_tuples.size() / dim), q_desc.data(),
                                   (int)(q_desc.size() / 4), q_max_edges));
    } else if (q_kind == Q_VIEW && !q_polys.empty()) {
        chk(lt_submit_object(g_ctx, nullptr, q_verts.data(), (int)(q_verts.size() / 4), q_polys.data(), q_aux.data(),
                             (int)(q_polys.size() / 4), q_pv.data(), (int)(q_pv.size() / 6), q_type, q_focus, q_force,
                             q_max_edges));
    } else if (q_kind == Q_OBJECTS && !q_polys.empty()) {
        chk(lt_submit_objects(g_ctx, q_objects.data(), (int)q_objects.size(), q_polyobj.data(), q_verts.data(),
                              (int)(q_verts.size() / 4), q_polys.data(), q_aux.data(), (int)(q_polys.size() / 4), q_pv.data(),
                              (int)(q_pv.size() / 6), q_type, q_focus, q_force, q_max_edges));
    } else if (q_kind == Q_LINES && !q_lines.empty()) {
        chk(lt_submit_lines(g_ctx, q_lines.data(), (int)(q_lines.size() / 8)));
    } else {
        submitted = false;
    }
    q_kind = Q_NONE;
    // The epoch only moves when a batch has gone out: a texture whose entry carries the current epoch
    // may be named by the open batch and keeps its handle and its texels until that batch is submitted.
    if (submitted) g_tex_epoch++;
    q_lines.clear();
    q_prp.clear();
    q_tuples.clear(); q_desc.clear(); q_verts.clear(); q_polys.clear(); q_aux.clear(); q_pv.clear();
    q_objects.clear(); q_polyobj.clear();
    q_max_edges = 0;
}

void open_queue(QKind kind, int type)
{
    if (q_kind != kind || q_type != type || ((kind == Q_VIEW || kind == Q_OBJECTS || type == LT_TEXTURED_PRP) && (q_focus != M_camera_log_focus || q_force != M_force_linear_tmapping)))
        run_queue();
    q_kind = kind; q_type = type; q_focus = M_camera_log_focus; q_force = M_force_linear_tmapping;
    g_host_frame_current = false;
}

// The object queues are written once and next read by the copy engine, not by this thread: 16-byte
// stores that go past the caches (no read of the line that is about to be overwritten, and the queue does
// not evict the program's own data).  run_queue fences them before the arrays are handed over.
inline void stream_ints(int32_t *dst, const int32_t *src, size_t ints)
{
    if ((((uintptr_t)dst) & 15) == 0 && (ints & 3) == 0) {
        for (size_t i = 0; i < ints; i += 4) _mm_stream_si128((__m128i *)(dst + i), _mm_loadu_si128((const __m128i *)(src + i)));
    } else {
        memcpy(dst, src, ints * sizeof(int32_t));
    }
}

template <class V>
void flatten_polygon(const M_polygon *p, int vertex_base, V &polys, V &aux, V &pv)
{
    int first = (int)(pv.size() / 6);
    int tex = -1;
    if (p->m_texture) tex = texture_handle(p->m_texture->m_texture_bytes, p->m_texture->m_log_texture_size);
    const int32_t hd[4] = {p->m_colour, tex, p->m_no_edges, first};
    polys.insert(polys.end(), hd, hd + 4);
    const int32_t ax[4] = {p->m_red, p->m_green, p->m_blue, (p->m_type & 0xff) | (p->m_log_texture_space_size << 8)};
    aux.insert(aux.end(), ax, ax + 4);
    for (int k = 0; k <= p->m_no_edges; k++) {
        const int *r = p->m_vertices + k * M_LNG_POLYGON_VERTEX;
        const int32_t rec[6] = {r[0] + vertex_base, r[1], r[2], r[3], r[4], r[5]};
        pv.insert(pv.end(), rec, rec + 6);
    }
}

}  // namespace

// ------------------------------------------------------------------ platform

__attribute__((weak)) void HW_error(char *s, ...)
{
    va_list lst;
    va_start(lst, s);
    vfprintf(stderr, s, lst);
    va_end(lst);
    fputc('\n', stderr);
    exit(1);                                         // the reference shows the message and exits
}

void HW_init_screen_size(int width, int height) { g_w = width; g_h = height; }
int HW_screen_x_size(void) { return g_w; }
int HW_screen_y_size(void) { return g_h; }
void HW_select_device(int device) { g_device = device; }

void HW_flush(void)
{
    need_ctx();
    run_queue();
    chk(lt_readback(g_ctx, G_c_buffer, G_z_buffer));
    g_host_frame_current = true;
}

// ------------------------------------------------------------------ colour (Colour.cpp:44-102)

void CL_init_colour(void) {}                         // the 256x256 table is evaluated arithmetically

HW_pixel CL_colour(int r, int g, int b)
{
    if (r > 255) r = 255;
    if (g > 255) g = 255;
    if (b > 255) b = 255;
    return (r << 16) | (g << 8) | b;
}

HW_pixel CL_light(HW_pixel colour, int r, int g, int b)
{
    int cr = (colour >> 16) & 0xff, cg = (colour >> 8) & 0xff, cb = colour & 0xff;
    if (r > 255) r = 255;
    if (g > 255) g = 255;
    if (b > 255) b = 255;
    return (((cr * r) / 255) << 16) | (((cg * g) / 255) << 8) | ((cb * b) / 255);
}

// ------------------------------------------------------------------ graphics

void G_init_graphics(void)
{
    if (g_ctx) return;
    if (lt_ctx_create(g_device, g_w, g_h, &g_ctx) != LT_OK) {
        char buf[600];
        snprintf(buf, sizeof buf, "(Graphics) %s", lt_last_error(nullptr));
        HW_error(buf);
    }
    G_c_buffer_size = (long)g_w * g_h;
    // the host copy of the frame in pinned memory where the library can provide it: HW_flush then costs the
    // link's time for 8 bytes a pixel and no host copy (the reference's arrays are the C heap's,
    // GraphicsBase.cpp:76, 86; neither is ever freed)
    auto frame_array = [](size_t bytes) {
        void *p = nullptr;
        if (lt_host_alloc(bytes, &p) != LT_OK) p = malloc(bytes);
        return p;
    };
    G_c_buffer = (HW_pixel *)frame_array(sizeof(HW_pixel) * G_c_buffer_size);
    G_z_buffer = (int *)frame_array(sizeof(int) * G_c_buffer_size);
    if (!G_c_buffer || !G_z_buffer) HW_error((char *)"(Graphics) Not enough memory.\n");
    C_init_clipping(0, 0, g_w - 1, g_h - 1);
}

void G_clear(HW_pixel colour, int red, int green, int blue)
{
    need_ctx();
    run_queue();                                     // earlier draws belong to the frame being replaced
    chk(lt_clear(g_ctx, CL_light(colour, red, green, blue)));
    g_host_frame_current = false;
}

static void track_fill(int *edges, int length, int xdim, int ydim);

static void queue_screen(int type, int dim, int *edges, int length, HW_pixel colour, int tex)
{
    need_ctx();
    if (length < 2 || length + 1 > G_MAX_POLYGON_VERTICES) HW_error((char *)"(Graphics) polygon outside 2..15 edges.\n");
    open_queue(Q_SCREEN, type);
    const int32_t d[4] = {(int32_t)(q_tuples.size() / dim), length, colour, tex};
    q_desc.insert(q_desc.end(), d, d + 4);
    q_tuples.insert(q_tuples.end(), edges, edges + (length + 1) * dim);
    if (length > q_max_edges) q_max_edges = length;
    if (g_track_stores) track_fill(edges, length, dim, type == LT_TEXTURED_PRP ? G_LNG_SHADED : dim);
}

// GraphicsPoly.cpp:520-684; M_camera_log_focus is read at the time of the call, as the
// reference's function reads the global while it draws
void G_prp_textured_polygon(int *edges, int length, int *O, int *u, int *v, HW_pixel *texture, int log_texture_size,
                            int log_texture_scale)
{
    need_ctx();
    open_queue(Q_SCREEN, LT_TEXTURED_PRP);           // first: what it submits must not find this texture marked as in use
    const int tex = texture_handle(texture, log_texture_size, true);
    queue_screen(LT_TEXTURED_PRP, G_LNG_TEXTURED, edges, length, 0, tex);
    const int32_t k[12] = {O[0], O[1], O[2], O[6], O[7], u[0], u[1], u[2], v[0], v[1], v[2], log_texture_scale};
    q_prp.insert(q_prp.end(), k, k + 12);
}

// GraphicsBase.cpp:165-253.  A clipped end point takes depth 0 (what the reference reads in
// a program that draws only wireframes; see lt_submit_lines in lt_gpu.h).
void G_line(int *vertex1, int *vertex2, HW_pixel colour, int red, int green, int blue)
{
    need_ctx();
    if (g_track_stores) {
        // GraphicsBase.cpp:191-193 on the host: the two clips with dimension 2 hand back pointers into
        // the clippers' static stores, whose third word is whatever earlier calls left there
        int *v1 = vertex1, *v2 = vertex2;
        if (!C_line_x_clipping(&v1, &v2, 2) || !C_line_y_clipping(&v1, &v2, 2)) return;
        open_queue(Q_LINES, 0);
        const int32_t l[8] = {v1[0], v1[1], v1[2], v2[0], v2[1], v2[2], CL_light(colour, red, green, blue), 3};
        q_lines.insert(q_lines.end(), l, l + 8);
        return;
    }
    open_queue(Q_LINES, 0);
    const int32_t l[8] = {vertex1[0], vertex1[1], vertex1[2], vertex2[0], vertex2[1], vertex2[2],
                          CL_light(colour, red, green, blue), 0};
    q_lines.insert(q_lines.end(), l, l + 8);
}

// GraphicsBase.cpp:117-153: single points, queued with the lines (draw order is kept)
void G_dot(int *vertex, HW_pixel colour, int red, int green, int blue)
{
    need_ctx();
    open_queue(Q_LINES, 0);
    const int32_t l[8] = {vertex[0], vertex[1], vertex[2], 0, 0, 0, CL_light(colour, red, green, blue), 1};
    q_lines.insert(q_lines.end(), l, l + 8);
}

void G_pixel(int *vertex, HW_pixel colour)
{
    need_ctx();
    open_queue(Q_LINES, 0);
    const int32_t l[8] = {vertex[0], vertex[1], 0, 0, 0, 0, colour, 2};
    q_lines.insert(q_lines.end(), l, l + 8);
}

void G_flat_polygon(int *edges, int length, HW_pixel colour, int red, int green, int blue)
{
    queue_screen(LT_FLAT, G_LNG_FLAT, edges, length, CL_light(colour, red, green, blue), 0);
}

void G_shaded_polygon(int *edges, int length, HW_pixel colour)
{
    queue_screen(LT_SHADED, G_LNG_SHADED, edges, length, colour, 0);
}

void G_lin_textured_polygon(int *edges, int length, HW_pixel *texture, int log_texture_size)
{
    need_ctx();
    open_queue(Q_SCREEN, LT_TEXTURED);
    const int tex = texture_handle(texture, log_texture_size, true);
    queue_screen(LT_TEXTURED, G_LNG_TEXTURED, edges, length, 0, tex);
}

// ------------------------------------------------------------------ transforms (TransBs.cpp / Trans.cpp)

void T_init_math(void)
{
    // Trans.cpp:62-77: scale, then truncate.  (The file the reference's project builds,
    // TransBs.cpp:70-71, truncates first, which zeroes the tables except cos[0].)
    for (int i = 0; i < 256; i++) {
        T_sin_tab[i] = (int)(sin(i / 40.7436642) * (1 << T_P));
        T_cos_tab[i] = (int)(cos(i / 40.7436642) * (1 << T_P));
    }
}

void T_set_self_rotation(unsigned char alp, unsigned char bet, unsigned char gam)
{
    const int ca = T_cos_tab[alp], sa = T_sin_tab[alp], cb = T_cos_tab[bet], sb = T_sin_tab[bet];
    const int cg = T_cos_tab[gam], sg = T_sin_tab[gam];
    const int sbsg = fx(sb, sg), sbcg = fx(sb, cg);
    Ts[0] = fx(ca, cg) - fx(sa, sbsg);   Ts[1] = fx(sa, cg) + fx(ca, sbsg);   Ts[2] = fx(cb, sg);     // sx1 sy1 sz1
    Ts[3] = -fx(sa, cb);                 Ts[4] = fx(ca, cb);                  Ts[5] = -sb;            // sx2 sy2 sz2
    Ts[6] = -fx(ca, sg) - fx(sa, sbcg);  Ts[7] = fx(ca, sbcg) - fx(sa, sg);   Ts[8] = fx(cb, cg);     // sx3 sy3 sz3
}

void T_set_world_rotation(unsigned char alp, unsigned char bet, unsigned char gam)
{
    const int ca = T_cos_tab[alp], sa = T_sin_tab[alp], cb = T_cos_tab[bet], sb = T_sin_tab[bet];
    const int cg = T_cos_tab[gam], sg = T_sin_tab[gam];
    const int sbsa = fx(sb, sa), sbca = fx(sb, ca);
    Tw[0] = fx(sg, sbsa) + fx(cg, ca);   Tw[1] = fx(cb, sa);   Tw[2] = fx(sg, ca) - fx(cg, sbsa);     // wx1 wy1 wz1
    Tw[3] = fx(sg, sbca) - fx(cg, sa);   Tw[4] = fx(cb, ca);   Tw[5] = -fx(cg, sbca) - fx(sg, sa);    // wx2 wy2 wz2
    Tw[6] = -fx(sg, cb);                 Tw[7] = sb;           Tw[8] = fx(cg, cb);                    // wx3 wy3 wz3
}

void T_cancatinate_self_world(int addx, int addy, int addz)
{
    for (int r = 0; r < 3; r++) {
        const int wx = Tw[3 * r], wy = Tw[3 * r + 1], wz = Tw[3 * r + 2];
        Tc[4 * r + 0] = wadd(wadd(fx(Ts[0], wx), fx(Ts[3], wy)), fx(Ts[6], wz));
        Tc[4 * r + 1] = wadd(wadd(fx(Ts[1], wx), fx(Ts[4], wy)), fx(Ts[7], wz));
        Tc[4 * r + 2] = wadd(wadd(fx(Ts[2], wx), fx(Ts[5], wy)), fx(Ts[8], wz));
        Tc[4 * r + 3] = wadd(wadd(fx(addx, wx), fx(addy, wy)), fx(addz, wz));
    }
}

// Small vertex arrays are transformed on the host with the same arithmetic; the batched
// GPU transform is what M_render_polygon_object uses (fused into its setup kernel).
void T_cancatinated_rotation(int *from, int *to, int length)
{
    for (int i = 0; i < length; i++, from += 4, to += 4) {
        const int x = from[0], y = from[1], z = from[2];
        for (int r = 0; r < 3; r++)
            to[r] = wadd(wadd(wadd(fx(Tc[4 * r], x), fx(Tc[4 * r + 1], y)), fx(Tc[4 * r + 2], z)), Tc[4 * r + 3]);
    }
}

// ------------------------------------------------------------------ clip window (Clipp2D.cpp:35-82)

// Clipp2D.cpp:36-77.  Draws queued so far keep the window they were issued under.  A window
// that is empty or leaves the screen cannot be handed to the GPU: C_set_bounds then reports
// "no clipping window" (its callers skip the drawing), C_init_clipping stops.
static void apply_window()
{
    if (!g_ctx) return;
    run_queue();
    chk(lt_set_clip_window(g_ctx, Cb[0], Cb[1], Cb[2], Cb[3]));
}

void C_init_clipping(int minx, int miny, int maxx, int maxy)
{
    if (g_ctx && (minx < 0 || miny < 0 || maxx > g_w - 1 || maxy > g_h - 1 || minx >= maxx || miny >= maxy))
        HW_error((char *)"(Clipper) the clip window must be a non-empty part of the screen.\n");
    Cb[0] = minx; Cb[1] = miny; Cb[2] = maxx; Cb[3] = maxy;
    apply_window();
}

void C_get_bounds(int *minx, int *miny, int *maxx, int *maxy)
{
    *minx = Cb[0]; *miny = Cb[1]; *maxx = Cb[2]; *maxy = Cb[3];
}

int C_set_bounds(int minx, int miny, int maxx, int maxy)
{
    if (minx > Cb[0]) Cb[0] = minx;
    if (maxx < Cb[2]) Cb[2] = maxx;
    if (miny > Cb[1]) Cb[1] = miny;
    if (maxy < Cb[3]) Cb[3] = maxy;
    if (Cb[0] >= Cb[2] || Cb[1] >= Cb[3]) return 0;       // no window: the caller draws nothing (EngBase.cpp:629)
    apply_window();
    return 1;
}

// ------------------------------------------------------------------ host-side clippers and projection
//
// The reference's callers may use these directly (Clipper.h:29-35, Trans.h:42,62); the GPU
// path has its own copies inside the set-up kernels.  M_set_gate below is built on them.

int C_2D_clipping = 0;                               // Clipp2D.cpp:22
int C_3D_clipping = 0;                               // Clipp3D.cpp

namespace {

#define LT_C_MAX_DIMENSIONS 8                        // Clipper.h:16
#define LT_C_Z_CLIPPING_MIN 10                       // Clipper.h:18

// One binary-search clipper behind the three line-clipping entry points.  The two ends are
// ordered along `axis`; an end at or beyond a bound is replaced by the point where halving
// the segment -- every dimension alike, floor of the mean -- lands exactly on the bound
// (Clipp2D.cpp:118-152, Clipp3D.cpp:91-104).  As in the reference the caller's pointer is
// redirected to one of the entry point's static stores, valid until its next call.
int clip_segment(int **vertex1, int **vertex2, int dimension, int axis, int bmin, int bmax, bool both,
                 int (*store)[LT_C_MAX_DIMENSIONS], int *first_moved)
{
    *first_moved = 0;
    const bool swapped = !((*vertex1)[axis] < (*vertex2)[axis]);
    int **lo = swapped ? vertex2 : vertex1, **hi = swapped ? vertex1 : vertex2;
    if (both ? ((*lo)[axis] >= bmax || (*hi)[axis] <= bmin) : ((*hi)[axis] <= bmin)) return 0;
    for (int side = 0; side < (both ? 2 : 1); side++) {
        const int bound = side == 0 ? bmin : bmax;
        if (side == 0 ? !((*lo)[axis] <= bmin) : !((*hi)[axis] >= bmax)) continue;
        // side 0 searches from the low end, side 1 from the high end
        int *m = store[3 * side + (side == 0 ? 0 : 1)], *l, *r;
        if (side == 0) { r = store[1]; l = store[2]; memcpy(m, *lo, sizeof(int) * dimension); memcpy(r, *hi, sizeof(int) * dimension); }
        else           { l = store[3]; r = store[5]; memcpy(l, *lo, sizeof(int) * dimension); memcpy(m, *hi, sizeof(int) * dimension); }
        bool below = side == 0;                      // is the search point on the low side of the bound?
        while (m[axis] != bound) {
            if (below) std::swap(l, m); else std::swap(r, m);
            for (int i = 0; i < dimension; i++) m[i] = wadd(l[i], r[i]) >> 1;
            below = m[axis] < bound;
        }
        if (side == 0) { *lo = m; *first_moved = swapped ? 0 : 1; }
        else           { *hi = m; *first_moved |= swapped ? 1 : 0; }
    }
    return 1;
}

// C_polygon_x_clipping / C_polygon_z_clipping (Clipp2D.cpp:247-275, Clipp3D.cpp:129-159): every
// edge through the line clipper; an edge whose first end moved contributes both ends, any
// other surviving edge its second end; the first tuple written closes the polygon.
template <typename F>
int clip_polygon_host(int *from, int *to, int dimension, int length, F line_clip, const int *moved_flag)
{
    int n = 0;
    int *out = to;
    for (int i = 0; i < length; i++, from += dimension) {
        int *v1 = from, *v2 = from + dimension;
        if (!line_clip(&v1, &v2, dimension)) continue;
        if (*moved_flag) { memcpy(out, v1, sizeof(int) * dimension); out += dimension; n++; }
        memcpy(out, v2, sizeof(int) * dimension); out += dimension; n++;
    }
    memcpy(out, to, sizeof(int) * dimension);
    return n;
}

}  // namespace

int C_line_x_clipping(int **vertex1, int **vertex2, int dimension)
{
    static int store[6][LT_C_MAX_DIMENSIONS];
    return clip_segment(vertex1, vertex2, dimension, 0, Cb[0], Cb[2], true, store, &C_2D_clipping);
}

int C_line_y_clipping(int **vertex1, int **vertex2, int dimension)
{
    static int store[6][LT_C_MAX_DIMENSIONS];
    return clip_segment(vertex1, vertex2, dimension, 1, Cb[1], Cb[3], true, store, &C_2D_clipping);
}

int C_line_z_clipping(int **vertex1, int **vertex2, int dimension)
{
    static int store[6][LT_C_MAX_DIMENSIONS];
    return clip_segment(vertex1, vertex2, dimension, 2, LT_C_Z_CLIPPING_MIN, 0, false, store, &C_3D_clipping);
}

int C_polygon_x_clipping(int *from, int *to, int dimension, int length)
{
    return clip_polygon_host(from, to, dimension, length, C_line_x_clipping, &C_2D_clipping);
}

int C_polygon_z_clipping(int *from, int *to, int dimension, int length)
{
    return clip_polygon_host(from, to, dimension, length, C_line_z_clipping, &C_3D_clipping);
}

// Clipp3D.cpp:46-58: 0 outside the view pyramid, -1 across the near plane, 1 in front of it
int C_volume_clipping(int *min, int *max)
{
    if (max[2] < min[0] || max[2] < min[1] || max[2] < -max[0] || max[2] < -max[1] || max[2] <= LT_C_Z_CLIPPING_MIN) return 0;
    return min[2] < LT_C_Z_CLIPPING_MIN ? -1 : 1;
}

// TransBs.cpp:416-434 (_Z_BUFFER_): X' = (X << log_focus) / Z + W/2, Y' likewise, Z' = -0x7ffff / Z,
// 64-bit intermediates truncated to int; the other values of a tuple are passed on.
void T_perspective(int *from, int *to, int dimension, int length, int log_focus)
{
    for (int i = 0; i < length; i++, from += dimension, to += dimension) {
        const long long z = from[2];
        to[0] = wadd((int)(((long long)from[0] * ((long long)1 << log_focus)) / z), g_w / 2);
        to[1] = wadd((int)(((long long)from[1] * ((long long)1 << log_focus)) / z), g_h / 2);
        to[2] = (int)(-0x7ffffLL / z);
        for (int j = 3; j < dimension; j++) to[j] = from[j];
    }
}

// TransLn.cpp:218-223
int T_normal_z_negative(int *from1, int *from2, int *from3)
{
    const long long a = (long long)(int)((unsigned)from2[0] - (unsigned)from1[0]) * (int)((unsigned)from3[1] - (unsigned)from1[1]);
    const long long b = (long long)(int)((unsigned)from2[1] - (unsigned)from1[1]) * (int)((unsigned)from3[0] - (unsigned)from1[0]);
    return a - b < 0;
}

// ------------------------------------------------------------------ engine (EngBase.cpp / EngPoly.cpp)

// ---- the clippers' static stores (HW_track_clipper_stores)
//
// G_line clips with dimension 2 (GraphicsBase.cpp:191-193), so the depth of a clipped end point
// is the third word of one of the 2-D clippers' static stores (Clipp2D.cpp:99-104, 172-177):
// a leftover of the last polygon clips.  With tracking on, every polygon that is drawn makes
// the reference's own sequence of clipper calls on the host as well (their results are
// dropped: the GPU clips for itself), so that the stores hold what the reference's would, and
// G_line clips on the host and reads them.

void HW_track_clipper_stores(int on)
{
    if (g_ctx) run_queue();
    g_track_stores = on != 0;
}

// G_*_polygon up to the edge scan: C_polygon_x_clipping, then GI_scan's y clip of every edge
// (GraphicsPoly.cpp:318-323, 136; the perspective mapper scans with the shaded dimension, :559)
static void track_fill(int *edges, int length, int xdim, int ydim)
{
    int out[64 * LT_C_MAX_DIMENSIONS];
    if (length + 1 > 16) return;
    const int n = C_polygon_x_clipping(edges, out, xdim, length);
    for (int i = 0; i < n && i < 62; i++) {
        int *v1 = out + i * xdim, *v2 = v1 + xdim;
        C_line_y_clipping(&v1, &v2, ydim);
    }
}

// M_render_polygon (EngBase.cpp:379-552) as far as its clipper calls go.  `wire`: draw the edges
// (the M_WIRE case is drawn from here when tracking is on, one G_line per edge).
static void track_view_polygon(const M_polygon *poly, const int *vertices, const int *colours, const int *textures, int type)
{
    const int e = poly->m_no_edges;
    if (e < 2 || e + 1 > 16) return;
    const bool textured = type == M_TEXTURED && poly->m_texture != nullptr;
    const int dim = (type == M_WIRE || type == M_FLAT) ? 3 : (textured ? 8 : 6);
    int orig[17 * 8], clip[64 * 8], persp[64 * 8];
    int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
    for (int k = 0; k <= e; k++) {                  // CI_construct_tuples, EngBase.cpp:47-138
        const int *r = poly->m_vertices + k * M_LNG_POLYGON_VERTEX;
        const int *v = vertices + (size_t)r[0] * T_LNG_VECTOR;
        int *t = orig + k * dim;
        for (int q = 0; q < 3; q++) { t[q] = v[q]; mn[q] = v[q] < mn[q] ? v[q] : mn[q]; mx[q] = v[q] > mx[q] ? v[q] : mx[q]; }
        if (dim >= 6) for (int q = 0; q < 3; q++) t[3 + q] = (int)((unsigned)(colours ? colours[(size_t)r[0] * 3 + q] : r[1 + q]) << 16);
        if (dim == 8) for (int q = 0; q < 2; q++) t[6 + q] = (int)((unsigned)(textures ? textures[(size_t)r[0] * 2 + q] : r[4 + q]) << 16);
    }
    const int cnd = C_volume_clipping(mn, mx);
    if (cnd == 0) return;
    int number = e;
    int *clipped = orig;
    if (cnd == -1) { number = C_polygon_z_clipping(orig, clip, dim, number); clipped = clip; }
    if (number + 1 > 60) return;
    T_perspective(clipped, persp, dim, number + 1, M_camera_log_focus);
    if (!T_normal_z_negative(persp, persp + dim, persp + 2 * dim)) return;
    if (type == M_WIRE) {
        for (int i = 0; i < number; i++) G_line(persp + i * dim, persp + (i + 1) * dim, poly->m_colour, poly->m_red, poly->m_green, poly->m_blue);
        return;
    }
    const bool prp = textured && (M_force_linear_tmapping || clipped[2] < G_Z_MAPPING_SWITCH);
    track_fill(persp, number, dim, prp ? G_LNG_SHADED : dim);
}

void M_init_rendering(int type) { g_rendering_type = type; }

void M_set_camera(unsigned char alp, unsigned char bet, unsigned char gam, int x, int y, int z, int log_focus)
{
    T_set_world_rotation((unsigned char)(256 - alp), (unsigned char)(256 - bet), (unsigned char)(256 - gam));
    M_camera_x = -x; M_camera_y = -y; M_camera_z = -z;
    M_camera_log_focus = log_focus;
}

static int lt_type()
{
    if (g_rendering_type != M_WIRE && g_rendering_type != M_FLAT && g_rendering_type != M_SHADED &&
        g_rendering_type != M_TEXTURED)
        HW_error((char *)"(Engine) no such rendering type.\n");
    return g_rendering_type;
}

// `vertices` are already transformed (the caller ran T_cancatinated_rotation); the
// arrays may be temporaries, so the vertices this polygon uses are copied now.
void M_render_polygon(struct M_polygon *poly, int *vertices, int *colours, int *textures)
{
    need_ctx();
    const int type = lt_type();
    if (g_track_stores) {
        track_view_polygon(poly, vertices, colours, textures, type);
        if (type == M_WIRE) return;                  // drawn as host-clipped lines
    }
    open_queue(Q_VIEW, type);
    // (before anything of this polygon is queued: a new texture may have to submit the open batch for a handle)
    if (poly->m_texture) texture_handle(poly->m_texture->m_texture_bytes, poly->m_texture->m_log_texture_size, true);
    open_queue(Q_VIEW, type);
    const int base = (int)(q_verts.size() / 4);
    M_polygon tmp = *poly;
    std::vector<int> recs((size_t)(poly->m_no_edges + 1) * M_LNG_POLYGON_VERTEX);
    for (int k = 0; k <= poly->m_no_edges; k++) {
        const int *r = poly->m_vertices + k * M_LNG_POLYGON_VERTEX;
        const int *v = vertices + (size_t)r[0] * T_LNG_VECTOR;
        const int32_t vv[4] = {v[0], v[1], v[2], 0};
        q_verts.insert(q_verts.end(), vv, vv + 4);
        memcpy(&recs[(size_t)k * 6], r, sizeof(int) * 6);
        // CI_construct_tuples, EngBase.cpp:106-131: with the optional arrays R,G,B and Tx,Ty come
        // from them, indexed like the vertices, and the record's own fields are skipped
        if (colours) for (int q = 0; q < 3; q++) recs[(size_t)k * 6 + 1 + q] = colours[(size_t)r[0] * 3 + q];
        if (textures) for (int q = 0; q < 2; q++) recs[(size_t)k * 6 + 4 + q] = textures[(size_t)r[0] * 2 + q];
        recs[(size_t)k * 6] = k;
    }
    tmp.m_vertices = recs.data();
    flatten_polygon(&tmp, base, q_polys, q_aux, q_pv);
    if (poly->m_no_edges > q_max_edges) q_max_edges = poly->m_no_edges;
}

// ------------------------------------------------------------------ bicubic patches (EngBcub.cpp)
//
// Tessellation is host work here as in the reference (at most 33 x 33 points a patch, halving
// integer control polygons); normals and shading run on the GPU (lt_plane_normals,
// lt_shade_points) and every square is drawn through M_render_polygon, i.e. the GPU queue.

namespace {

inline int half_sum(int a, int b) { return wadd(a, b) / 2; }       // (a + b) / 2 of the reference: truncating

// Points of the cubic Bezier curve with controls b1..b4 at `length` = 2^k + 1 equally spaced
// parameters, `stride` ints apart (EngBcub.cpp:113-152): the ends are the outer controls, the
// middle comes from three rounds of halving, and each half is a curve again.
void curve_points(const int *b1, const int *b2, const int *b3, const int *b4, int *points, int stride, int length, bool top)
{
    if (top) {
        memcpy(points, b1, sizeof(int) * 3);
        memcpy(points + (size_t)(length - 1) * stride, b4, sizeof(int) * 3);
    }
    if (length <= 2) return;
    int b12[3], b23[3], b34[3], b123[3], b234[3];
    int *mid = points + (size_t)(length / 2) * stride;
    for (int k = 0; k < 3; k++) {
        b12[k] = half_sum(b1[k], b2[k]); b23[k] = half_sum(b2[k], b3[k]); b34[k] = half_sum(b3[k], b4[k]);
        b123[k] = half_sum(b12[k], b23[k]); b234[k] = half_sum(b23[k], b34[k]);
        mid[k] = half_sum(b123[k], b234[k]);
    }
    curve_points(b1, b12, b123, mid, points, stride, length / 2 + 1, false);
    curve_points(mid, b234, b34, b4, mid, stride, length / 2 + 1, false);
}

// Tangents at the same parameters (EngBcub.cpp:163-196): differences of neighbouring controls of
// the halved polygons.
void curve_tangents(const int *b1, const int *b2, const int *b3, const int *b4, int *tangents, int stride, int length, bool top)
{
    if (top)
        for (int k = 0; k < 3; k++) {
            tangents[k] = (int)((unsigned)b2[k] - (unsigned)b1[k]);
            tangents[(size_t)(length - 1) * stride + k] = (int)((unsigned)b4[k] - (unsigned)b3[k]);
        }
    if (length <= 2) return;
    int b12[3], b23[3], b34[3], b123[3], b234[3], b1234[3];
    int *mid = tangents + (size_t)(length / 2) * stride;
    for (int k = 0; k < 3; k++) {
        b12[k] = half_sum(b1[k], b2[k]); b23[k] = half_sum(b2[k], b3[k]); b34[k] = half_sum(b3[k], b4[k]);
        b123[k] = half_sum(b12[k], b23[k]); b234[k] = half_sum(b23[k], b34[k]);
        mid[k] = (int)((unsigned)b234[k] - (unsigned)b123[k]);
        b1234[k] = half_sum(b123[k], b234[k]);
    }
    curve_tangents(b1, b12, b123, b1234, tangents, stride, length / 2 + 1, false);
    curve_tangents(b1234, b234, b34, b4, mid, stride, length / 2 + 1, false);
}

// size x size points of a patch: four curves along one parameter through controls
// (k, k+4, k+8, k+12), then one curve per position along the other
void patch_points(const int *c, int *points, int size)
{
    std::vector<int> t((size_t)4 * size * T_LNG_VECTOR, 0);
    for (int k = 0; k < 4; k++)
        curve_points(c + k * T_LNG_VECTOR, c + (k + 4) * T_LNG_VECTOR, c + (k + 8) * T_LNG_VECTOR, c + (k + 12) * T_LNG_VECTOR,
                     &t[(size_t)k * size * T_LNG_VECTOR], T_LNG_VECTOR, size, true);
    for (int i = 0; i < size; i++)
        curve_points(&t[(size_t)i * T_LNG_VECTOR], &t[(size_t)(size + i) * T_LNG_VECTOR], &t[(size_t)(2 * size + i) * T_LNG_VECTOR],
                     &t[(size_t)(3 * size + i) * T_LNG_VECTOR], points + (size_t)i * size * T_LNG_VECTOR, T_LNG_VECTOR, size, true);
}

int patch_size(const M_bicubic *b, bool init)
{
    const int size = (1 << b->m_log_render_size) + 1;
    if (b->m_log_render_size < 0 || (init ? size >= M_MAX_RENDER_SIZE : size > M_MAX_RENDER_SIZE))
        HW_error((char *)"(Engine) Can't subdivide a bicubic this much.\n");       // EngBcub.cpp:215-216, 287-288
    return size;
}

int M_square_v[30] = {2, 0, 0, 0, 0, 0,  3, 0, 0, 0, 0, 0,  1, 0, 0, 0, 0, 0,  0, 0, 0, 0, 0, 0,  2, 0, 0, 0, 0, 0};
M_polygon M_square = {M_QUAD_XY, 0, 0, 0, 0, 0, nullptr, 4, M_square_v, nullptr};   // EngBcub.cpp:49-60

}  // namespace

// EngBcub.cpp:204-265: the normal at every point of the tessellation = vector product of the
// tangents along the two parameters (T_normal_vectors, i.e. T_normal_plane from the origin:
// computed by lt_plane_normals).  A zero-length product stops the reference and this.
void M_init_bicubic(struct M_bicubic *bicubic)
{
    need_ctx();
    const int size = patch_size(bicubic, true), n = size * size;
    const int *c = bicubic->m_controls;
    bicubic->m_normals = (int *)malloc(sizeof(int) * T_LNG_VECTOR * n);
    bicubic->m_intensities = (int *)malloc(sizeof(int) * 3 * n);
    if (!bicubic->m_normals || !bicubic->m_intensities) HW_error((char *)"(Engine) Not enough memory.\n");
    // vertex 0 is the origin, then the two tangent fields
    std::vector<int32_t> v((size_t)(1 + 2 * n) * 4, 0), t((size_t)4 * size * T_LNG_VECTOR, 0);
    int *tan1 = &v[4], *tan2 = &v[(size_t)(1 + n) * 4];
    for (int k = 0; k < 4; k++)
        curve_points(c + 4 * k * T_LNG_VECTOR, c + (4 * k + 1) * T_LNG_VECTOR, c + (4 * k + 2) * T_LNG_VECTOR,
                     c + (4 * k + 3) * T_LNG_VECTOR, &t[(size_t)k * size * T_LNG_VECTOR], T_LNG_VECTOR, size, true);
    for (int i = 0; i < size; i++)
        curve_tangents(&t[(size_t)i * T_LNG_VECTOR], &t[(size_t)(size + i) * T_LNG_VECTOR], &t[(size_t)(2 * size + i) * T_LNG_VECTOR],
                       &t[(size_t)(3 * size + i) * T_LNG_VECTOR], tan1 + i * T_LNG_VECTOR, T_LNG_VECTOR * size, size, true);
    for (int k = 0; k < 4; k++)
        curve_points(c + k * T_LNG_VECTOR, c + (k + 4) * T_LNG_VECTOR, c + (k + 8) * T_LNG_VECTOR, c + (k + 12) * T_LNG_VECTOR,
                     &t[(size_t)k * size * T_LNG_VECTOR], T_LNG_VECTOR, size, true);
    for (int i = 0; i < size; i++)
        curve_tangents(&t[(size_t)i * T_LNG_VECTOR], &t[(size_t)(size + i) * T_LNG_VECTOR], &t[(size_t)(2 * size + i) * T_LNG_VECTOR],
                       &t[(size_t)(3 * size + i) * T_LNG_VECTOR], tan2 + (size_t)i * size * T_LNG_VECTOR, T_LNG_VECTOR, size, true);
    std::vector<int32_t> polys((size_t)n * 4), pv((size_t)n * 3 * 6, 0), out((size_t)n * 4);
    for (int i = 0; i < n; i++) {
        polys[(size_t)i * 4] = 0; polys[(size_t)i * 4 + 1] = -1; polys[(size_t)i * 4 + 2] = 2; polys[(size_t)i * 4 + 3] = 3 * i;
        pv[(size_t)(3 * i) * 6] = 0; pv[(size_t)(3 * i + 1) * 6] = 1 + i; pv[(size_t)(3 * i + 2) * 6] = 1 + n + i;
    }
    chk(lt_plane_normals(g_ctx, v.data(), 1 + 2 * n, polys.data(), n, pv.data(), 3 * n, out.data()));
    memcpy(bicubic->m_normals, out.data(), sizeof(int) * 4 * n);
}

// EngBcub.cpp:271-326: controls to world space, points from them, M_shade_vertex at every point
// with its rotated normal (lt_shade_points)
void M_shade_bicubic(struct M_bicubic *bicubic, int x, int y, int z, int alp, int bet, int gam, int no_lights,
                     struct M_light **lights)
{
    need_ctx();
    const int size = patch_size(bicubic, false), n = size * size;
    if (no_lights > LT_MAX_LIGHTS) HW_error((char *)"(Engine) more lights than the GPU path takes.\n");
    T_set_self_rotation((unsigned char)alp, (unsigned char)bet, (unsigned char)gam);
    int c[16 * T_LNG_VECTOR];
    for (int i = 0; i < 16; i++) {
        const int *p = bicubic->m_controls + i * T_LNG_VECTOR;
        for (int r = 0; r < 3; r++)
            c[i * T_LNG_VECTOR + r] = wadd(wadd(fx(Ts[3 * r], p[0]), fx(Ts[3 * r + 1], p[1])), fx(Ts[3 * r + 2], p[2]));
        c[i * T_LNG_VECTOR] = wadd(c[i * T_LNG_VECTOR], x);
        c[i * T_LNG_VECTOR + 1] = wadd(c[i * T_LNG_VECTOR + 1], y);
        c[i * T_LNG_VECTOR + 2] = wadd(c[i * T_LNG_VECTOR + 2], z);
        c[i * T_LNG_VECTOR + 3] = 0;
    }
    std::vector<int32_t> points((size_t)n * 4, 0), rgb((size_t)n * 4, 0);
    patch_points(c, points.data(), size);
    std::vector<lt_light> ls((size_t)(no_lights > 0 ? no_lights : 0));
    for (int j = 0; j < no_lights; j++) {
        ls[j].type = lights[j]->m_type;
        for (int k = 0; k < 4; k++) ls[j].param[k] = lights[j]->m_parameter[k];
        ls[j].red = lights[j]->m_red; ls[j].green = lights[j]->m_green; ls[j].blue = lights[j]->m_blue;
    }
    chk(lt_shade_points(g_ctx, Ts, points.data(), bicubic->m_normals, n, ls.data(), no_lights, rgb.data()));
    for (int i = 0; i < n; i++)
        for (int q = 0; q < 3; q++) bicubic->m_intensities[i * 3 + q] = rgb[(size_t)i * 4 + q];
}

// EngBcub.cpp:332-403 with a depth buffer (no sorting): controls to view space with M_PB more
// fraction bits, points, a grid of texture coordinates, then every square as a four-edge
// polygon through M_render_polygon (MI_render_square, :68-102)
void M_render_bicubic(struct M_bicubic *bicubic, int x, int y, int z, int alp, int bet, int gam)
{
    need_ctx();
    const int size = patch_size(bicubic, false), n = size * size;
    T_set_self_rotation((unsigned char)alp, (unsigned char)bet, (unsigned char)gam);
    T_cancatinate_self_world(x + M_camera_x, y + M_camera_y, z + M_camera_z);
    int c[16 * T_LNG_VECTOR] = {0};
    T_cancatinated_rotation(bicubic->m_controls, c, 16);
    int incr = bicubic->m_log_texture_space_size - bicubic->m_log_render_size;
    if (incr <= 0) HW_error((char *)"(Engine) Too small a texture to fit onto a bicubic.\n");
    M_square.m_colour = bicubic->m_colour;
    M_square.m_log_texture_space_size = incr;
    M_square.m_texture = bicubic->m_texture;
    incr = 1 << incr;
    const int max = 1 << bicubic->m_log_texture_space_size;
    std::vector<int> points((size_t)n * T_LNG_VECTOR, 0), textures((size_t)n * 2);
    for (int i = 0; i < size; i++)
        for (int j = 0; j < size; j++) { textures[(size_t)(i * size + j) * 2] = j * incr; textures[(size_t)(i * size + j) * 2 + 1] = max - i * incr; }
    for (int i = 0; i < 16; i++)
        for (int q = 0; q < 3; q++) c[i * T_LNG_VECTOR + q] = (int)((unsigned)c[i * T_LNG_VECTOR + q] << M_PB);
    patch_points(c, points.data(), size);
    for (int i = 0; i < n; i++) {
        if (i % size == size - 1 || i / size == size - 1) continue;
        const int no[4] = {i, i + 1, i + size, i + size + 1};
        int vert[4 * T_LNG_VECTOR], colr[4 * 3], tex[4 * 2];
        for (int k = 0; k < 4; k++) {
            for (int q = 0; q < 3; q++) vert[k * T_LNG_VECTOR + q] = points[(size_t)no[k] * T_LNG_VECTOR + q] >> M_PB;
            vert[k * T_LNG_VECTOR + 3] = 0;
            for (int q = 0; q < 3; q++) colr[k * 3 + q] = bicubic->m_intensities[no[k] * 3 + q];
            tex[k * 2] = textures[(size_t)no[k] * 2]; tex[k * 2 + 1] = textures[(size_t)no[k] * 2 + 1];
        }
        M_square.m_red = colr[0]; M_square.m_green = colr[1]; M_square.m_blue = colr[2];
        M_render_polygon(&M_square, vert, colr, tex);
    }
}

// EngBcub.cpp:410-488
void M_init_bicubic_object(struct M_bicubic_object *object)
{
    object->m_centres = (int *)malloc(sizeof(int) * T_LNG_VECTOR * (object->m_no_patches > 0 ? object->m_no_patches : 1));
    if (!object->m_centres) HW_error((char *)"(Engine) Not enough memory.\n");
    for (int i = 0; i < object->m_no_patches; i++) {
        M_init_bicubic(object->m_patches[i]);
        for (int q = 0; q < 3; q++) {
            int sum = 0;
            for (int j = 0; j < 16; j++) sum = wadd(sum, object->m_patches[i]->m_controls[j * T_LNG_VECTOR + q]);
            object->m_centres[i * T_LNG_VECTOR + q] = sum / 16;
        }
    }
}

void M_shade_bicubic_object(struct M_bicubic_object *object, int x, int y, int z, int alp, int bet, int gam, int no_lights,
                            struct M_light **lights)
{
    for (int i = 0; i < object->m_no_patches; i++) M_shade_bicubic(object->m_patches[i], x, y, z, alp, bet, gam, no_lights, lights);
}

void M_render_bicubic_object(struct M_bicubic_object *object, int x, int y, int z, int alp, int bet, int gam)
{
    if (object->m_no_patches > M_MAX_PATCHES) HW_error((char *)"(Engine) Too many bicubic patches per object.\n");
    const int old = M_force_linear_tmapping;
    M_force_linear_tmapping = 1;                     // EngBcub.cpp:469-470: the squares are small enough
    for (int i = 0; i < object->m_no_patches; i++) M_render_bicubic(object->m_patches[i], x, y, z, alp, bet, gam);
    M_force_linear_tmapping = old;
}

// EngBase.cpp:567-629: the clip window becomes its intersection with the screen extent of a
// portal polygon (view-space vertices, closed index list); 0 if the gate is behind the
// viewer, faces away or leaves no window.
int M_set_gate(struct M_gate *gate, int *vertices)
{
    int original1[G_MAX_POLYGON_VERTICES * LT_C_MAX_DIMENSIONS], original2[G_MAX_POLYGON_VERTICES * LT_C_MAX_DIMENSIONS];
    int perspective[G_MAX_POLYGON_VERTICES * LT_C_MAX_DIMENSIONS];
    int number = gate->m_no_edges;
    if (number + 1 > G_MAX_POLYGON_VERTICES) HW_error((char *)"(Engine) a gate of more than 15 edges.\n");
    int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
    for (int i = 0; i <= number; i++)
        for (int j = 0; j < 3; j++) {
            const int v = vertices[gate->m_vertices[i] * T_LNG_VECTOR + j];
            original1[i * 3 + j] = v;
            if (v > mx[j]) mx[j] = v;
            if (v < mn[j]) mn[j] = v;
        }
    const int cnd = C_volume_clipping(mn, mx);
    if (cnd == 0) return 0;
    int *clipped = original1;
    if (cnd == -1) { number = C_polygon_z_clipping(original1, original2, 3, number); clipped = original2; }
    T_perspective(clipped, perspective, 3, number + 1, M_camera_log_focus);
    if (!T_normal_z_negative(perspective, perspective + 3, perspective + 6)) return 0;
    int nminx = INT_MAX, nminy = INT_MAX, nmaxx = INT_MIN, nmaxy = INT_MIN;
    for (int i = 0; i < number; i++) {
        const int x = perspective[i * 3], y = perspective[i * 3 + 1];
        if (x > nmaxx) nmaxx = x;
        if (x < nminx) nminx = x;
        if (y > nmaxy) nmaxy = y;
        if (y < nminy) nminy = y;
    }
    return C_set_bounds(nminx, nminy, nmaxx, nmaxy);
}

// The list of normals under construction: MI_add_tuple (EngPoly.cpp:46-63) compares X Y Z
// and appends what it has not seen.
static int add_tuple(std::vector<int> &list, const int *t)
{
    const int n = (int)(list.size() / T_LNG_VECTOR);
    for (int i = 0; i < n; i++)
        if (list[(size_t)i * T_LNG_VECTOR] == t[0] && list[(size_t)i * T_LNG_VECTOR + 1] == t[1] &&
            list[(size_t)i * T_LNG_VECTOR + 2] == t[2])
            return i;
    if (n >= M_MAX_OBJECT_VERTICES) HW_error((char *)"(Engine) Not enough internal storage for vertices.\n");
    const int rec[T_LNG_VECTOR] = {t[0], t[1], t[2], 0};
    list.insert(list.end(), rec, rec + T_LNG_VECTOR);
    return n;
}

// EngPoly.cpp:355-459 with a depth buffer (no BSP ordering): the plane normal of every
// polygon is computed on the GPU (lt_plane_normals); the list handling -- planar polygons
// share one index per distinct normal, the vertices of curved polygons get the truncated
// mean of the plane normals of the curved polygons that use them -- follows the
// reference's order of insertion, so the indices come out the same.  m_normals arrays are
// malloc'ed as in the reference (the caller owns them).
void M_init_polygon_object(struct M_polygon_object *object)
{
    need_ctx();
    const int np = object->m_no_polygons;
    std::vector<int32_t> polys, pv;
    for (int i = 0; i < np; i++) {
        const M_polygon *p = object->m_polygons[i];
        const int32_t hd[4] = {p->m_colour, -1, p->m_no_edges, (int32_t)(pv.size() / 6)};
        polys.insert(polys.end(), hd, hd + 4);
        pv.insert(pv.end(), p->m_vertices, p->m_vertices + (size_t)(p->m_no_edges + 1) * M_LNG_POLYGON_VERTEX);
    }
    std::vector<int32_t> plane((size_t)(np > 0 ? np : 1) * 4);
    if (np > 0)     // a zero-length normal stops the reference (TransLn.cpp:166); chk() stops here with the library's message
        chk(lt_plane_normals(g_ctx, object->m_vertices, object->m_no_vertices, polys.data(), np, pv.data(), (int)(pv.size() / 6),
                             plane.data()));
    std::vector<int> list;
    for (int i = 0; i < np; i++) {
        M_polygon *p = object->m_polygons[i];
        if (p->m_type & M_PLANNAR) {
            p->m_normals = (int *)malloc(sizeof(int) * (p->m_no_edges + 1));
            if (p->m_normals == nullptr) HW_error((char *)"(Engine) Not enough memory.\n");
            const int idx = add_tuple(list, &plane[(size_t)i * 4]);
            for (int j = 0; j <= p->m_no_edges; j++) p->m_normals[j] = idx;
        }
    }
    for (int i = 0; i < object->m_no_vertices; i++) {
        int n[3] = {0, 0, 0}, num = 0;
        for (int j = 0; j < np; j++) {
            M_polygon *p = object->m_polygons[j];
            if (!(p->m_type & M_CURVED)) continue;
            if (p->m_normals == nullptr) {
                p->m_normals = (int *)calloc((size_t)p->m_no_edges + 1, sizeof(int));
                if (p->m_normals == nullptr) HW_error((char *)"(Engine) Not enough memory.\n");
            }
            for (int k = 0; k <= p->m_no_edges; k++)
                if (p->m_vertices[k * M_LNG_POLYGON_VERTEX] == i) {
                    p->m_normals[k] = -1;
                    for (int c = 0; c < 3; c++) n[c] = (int)((unsigned)n[c] + (unsigned)plane[(size_t)j * 4 + c]);
                    num++;
                }
        }
        if (num != 0) {
            n[0] /= num; n[1] /= num; n[2] /= num;
            const int idx = add_tuple(list, n);
            for (int j = 0; j < np; j++) {
                M_polygon *p = object->m_polygons[j];
                if (!(p->m_type & M_CURVED)) continue;
                for (int k = 0; k <= p->m_no_edges; k++)
                    if (p->m_normals[k] == -1) p->m_normals[k] = idx;
            }
        }
    }
    object->m_no_normals = (int)(list.size() / T_LNG_VECTOR);
    object->m_normals = (int *)malloc(sizeof(int) * (list.size() ? list.size() : 1));
    if (object->m_normals == nullptr) HW_error((char *)"(Engine) Not enough memory.\n");
    memcpy(object->m_normals, list.data(), sizeof(int) * list.size());
}

void M_shade_polygon_object(struct M_polygon_object *object, int x, int y, int z, int alp, int bet, int gam,
                            int no_lights, struct M_light **lights)
{
    need_ctx();
    if (object->m_no_vertices > M_MAX_OBJECT_VERTICES || object->m_no_normals > M_MAX_OBJECT_VERTICES)
        HW_error((char *)"(Engine) Can't handle this many vertices or normals in polygons.");    // EngPoly.cpp:481-484
    if (no_lights > LT_MAX_LIGHTS) HW_error((char *)"(Engine) more lights than the GPU path takes.\n");
    T_set_self_rotation((unsigned char)alp, (unsigned char)bet, (unsigned char)gam);
    std::vector<int32_t> polys, aux, pv, pn;
    for (int i = 0; i < object->m_no_polygons; i++) {
        const M_polygon *p = object->m_polygons[i];
        const int32_t hd[4] = {p->m_colour, -1, p->m_no_edges, (int32_t)(pv.size() / 6)};
        polys.insert(polys.end(), hd, hd + 4);
        const int32_t ax[4] = {p->m_red, p->m_green, p->m_blue, 0};
        aux.insert(aux.end(), ax, ax + 4);
        pv.insert(pv.end(), p->m_vertices, p->m_vertices + (size_t)(p->m_no_edges + 1) * M_LNG_POLYGON_VERTEX);
        pn.insert(pn.end(), p->m_normals, p->m_normals + (p->m_no_edges + 1));
    }
    std::vector<lt_light> ls((size_t)(no_lights > 0 ? no_lights : 0));
    for (int j = 0; j < no_lights; j++) {
        ls[j].type = lights[j]->m_type;
        for (int k = 0; k < 4; k++) ls[j].param[k] = lights[j]->m_parameter[k];
        ls[j].red = lights[j]->m_red; ls[j].green = lights[j]->m_green; ls[j].blue = lights[j]->m_blue;
    }
    chk(lt_shade_object(g_ctx, Ts, x, y, z, object->m_vertices, object->m_no_vertices, object->m_normals,
                        object->m_no_normals, polys.data(), object->m_no_polygons, pv.data(), pn.data(),
                        (int)(pv.size() / 6), aux.data(), ls.data(), no_lights));
    chk(lt_flush(g_ctx));          // a point light on a vertex stops the reference; HW_error says so here
    size_t r = 0;
    for (int i = 0; i < object->m_no_polygons; i++) {
        M_polygon *p = object->m_polygons[i];
        for (int k = 0; k <= p->m_no_edges; k++, r++) {
            int *rec = p->m_vertices + k * M_LNG_POLYGON_VERTEX;
            rec[1] = pv[r * 6 + 1]; rec[2] = pv[r * 6 + 2]; rec[3] = pv[r * 6 + 3];
        }
        p->m_red = aux[(size_t)i * 4]; p->m_green = aux[(size_t)i * 4 + 1]; p->m_blue = aux[(size_t)i * 4 + 2];
    }
}

// Objects are queued, not submitted: the reference's limits (fewer than 1024 vertices and at most
// 256 polygons an object, Engine.h:432-433) make a world of many triangles a sequence of many
// calls, and consecutive calls of one render type become one table of objects, one GPU batch
// (lt_submit_objects) -- the frame is that of the calls in turn.  The object's arrays are copied
// now (the caller may change them after the call returns, as the reference allows); the batch
// goes out when the frame is observed or cleared, when a draw of another kind follows, or when
// the render type, the focus or M_force_linear_tmapping change.
void M_render_polygon_object(struct M_polygon_object *object, int x, int y, int z, int alp, int bet, int gam)
{
    need_ctx();
    if (object->m_no_vertices >= M_MAX_OBJECT_VERTICES)
        HW_error((char *)"(Engine) Can't handle this many vertices in polygons.");      // EngPoly.cpp:507-508
    if (object->m_order) HW_error((char *)"(Engine) BSP-ordered objects are not supported by the GPU path.\n");
    const int type = lt_type();
    T_set_self_rotation((unsigned char)alp, (unsigned char)bet, (unsigned char)gam);
    T_cancatinate_self_world(x + M_camera_x, y + M_camera_y, z + M_camera_z);
    if (g_track_stores) {
        // the transform on the host too, for the clipper calls of every polygon (EngPoly.cpp:512-523);
        // a wireframe object is drawn from there, as host-clipped lines
        std::vector<int> tmp((size_t)object->m_no_vertices * T_LNG_VECTOR);
        T_cancatinated_rotation(object->m_vertices, tmp.data(), object->m_no_vertices);
        for (int i = 0; i < object->m_no_polygons; i++) track_view_polygon(object->m_polygons[i], tmp.data(), nullptr, nullptr, type);
        if (type == M_WIRE) return;
    }
    if (type == M_WIRE) {
        // wireframes read the depth buffer without writing it and are drawn object by object
        run_queue();
        g_host_frame_current = false;
        IVec polys, aux, pv;
        int me = 0;
        for (int i = 0; i < object->m_no_polygons; i++) {
            flatten_polygon(object->m_polygons[i], 0, polys, aux, pv);
            if (object->m_polygons[i]->m_no_edges > me) me = object->m_polygons[i]->m_no_edges;
        }
        chk(lt_submit_object(g_ctx, Tc, object->m_vertices, object->m_no_vertices, polys.data(), aux.data(),
                             object->m_no_polygons, pv.data(), (int)(pv.size() / 6), type, M_camera_log_focus,
                             M_force_linear_tmapping, me));
        g_tex_epoch++;                               // a batch has gone out, as in run_queue
        return;
    }
    open_queue(Q_OBJECTS, type);
    // Textures first, before anything of this object is queued: a new texture may have to submit the
    // open batch to free a handle.  If that happened part-way, the textures seen before it carry the
    // old batch's mark: look them all up once more (the open batch is empty now).
    for (int pass = 0; pass < 2; pass++) {
        const unsigned epoch = g_tex_epoch;
        for (int i = 0; i < object->m_no_polygons; i++) {
            const M_polygon *p = object->m_polygons[i];
            if (p->m_texture) texture_handle(p->m_texture->m_texture_bytes, p->m_texture->m_log_texture_size, true);
        }
        if (g_tex_epoch == epoch) break;
    }
    open_queue(Q_OBJECTS, type);
    lt_object o{};
    for (int k = 0; k < 12; k++) o.m[k] = Tc[k];
    o.vert_base = (int32_t)(q_verts.size() / 4);
    o.vert_count = object->m_no_vertices;
    const int index = (int)q_objects.size();
    q_objects.push_back(o);
    {
        const size_t at = q_verts.size(), more = (size_t)object->m_no_vertices * T_LNG_VECTOR;
        q_verts.resize(at + more);
        stream_ints(q_verts.data() + at, object->m_vertices, more);
    }
    // the object's polygons in one go: the arrays grow once, and every polygon is two short rows and a copy of its
    // vertex records (a world of thousands of small objects makes this loop the frame's host time)
    const int n = object->m_no_polygons;
    size_t nrec = 0;
    for (int i = 0; i < n; i++) nrec += (size_t)object->m_polygons[i]->m_no_edges + 1;
    const size_t p0 = q_polys.size(), a0 = q_aux.size(), v0 = q_pv.size(), o0 = q_polyobj.size();
    q_polys.resize(p0 + (size_t)n * 4); q_aux.resize(a0 + (size_t)n * 4); q_pv.resize(v0 + nrec * 6); q_polyobj.resize(o0 + (size_t)n);
    int32_t *pp = q_polys.data() + p0, *pa = q_aux.data() + a0, *pr = q_pv.data() + v0, *po = q_polyobj.data() + o0;
    int first = (int)(v0 / 6);
    for (int i = 0; i < n; i++) {
        const M_polygon *p = object->m_polygons[i];
        const int ne = p->m_no_edges;
        const int32_t hd[4] = {p->m_colour, p->m_texture ? texture_handle(p->m_texture->m_texture_bytes, p->m_texture->m_log_texture_size) : -1,
                               ne, first};
        const int32_t ax[4] = {p->m_red, p->m_green, p->m_blue, (p->m_type & 0xff) | (p->m_log_texture_space_size << 8)};
        stream_ints(pp, hd, 4);
        stream_ints(pa, ax, 4);
        stream_ints(pr, p->m_vertices, (size_t)(ne + 1) * M_LNG_POLYGON_VERTEX);
        pp += 4; pa += 4; pr += (size_t)(ne + 1) * M_LNG_POLYGON_VERTEX;
        first += ne + 1;
        po[i] = index;
        if (ne > q_max_edges) q_max_edges = ne;
    }
}
