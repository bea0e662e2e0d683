// extern "C" test shim over the host mirror (lt_engine.h), so that Python tests can drive
// the reference-named C++ API: it builds the M_polygon / M_polygon_object structures from
// the flat scene arrays and calls the mirror exactly as a reference-style program would.
#include "lt_engine.h"
#include "lt_data.h"
#include <stddef.h>
#include <stdlib.h>
#include <time.h>
#include <vector>

static M_texture g_texs[64];        // textures registered through lte_set_texture

extern "C" {

void lte_init(int w, int h)
{
    HW_init_screen_size(w, h);
    T_init_math();
    CL_init_colour();
    G_init_graphics();
}

void *lte_d_data(const char *name) { return D_data((char *)name); }

int *lte_cbuf(void) { return (int *)G_c_buffer; }
int *lte_zbuf(void) { return G_z_buffer; }
void lte_clear(int colour, int r, int g, int b) { G_clear(colour, r, g, b); }
void lte_flush(void) { HW_flush(); }
int lte_cl_light(int c, int r, int g, int b) { return CL_light(c, r, g, b); }
void lte_init_clipping(int minx, int miny, int maxx, int maxy) { C_init_clipping(minx, miny, maxx, maxy); }
int lte_set_bounds(int minx, int miny, int maxx, int maxy) { return C_set_bounds(minx, miny, maxx, maxy); }
// M_init_bicubic + M_shade_bicubic + M_render_bicubic (through a one-patch M_bicubic_object when
// as_object) of one patch into the current frame; normals[size*size][4], intensities[size*size][3]
void lte_bicubic(const int *controls, int log_render_size, int colour, int log_texture_space_size, int tex, int x, int y, int z,
                 int alp, int bet, int gam, const int *lights, int nl, int type, int log_focus, int as_object, int *normals,
                 int *intensities)
{
    M_bicubic b;
    M_bicubic *bp = &b;
    const int size = (1 << log_render_size) + 1;
    std::vector<M_light> ls(nl);
    std::vector<M_light *> lp(nl);
    for (int j = 0; j < nl; j++) {
        ls[j].m_type = lights[j * 8];
        for (int k = 0; k < 4; k++) ls[j].m_parameter[k] = lights[j * 8 + 1 + k];
        ls[j].m_red = lights[j * 8 + 5]; ls[j].m_green = lights[j * 8 + 6]; ls[j].m_blue = lights[j * 8 + 7];
        lp[j] = &ls[j];
    }
    b.m_log_render_size = log_render_size;
    b.m_colour = colour;
    b.m_log_texture_space_size = log_texture_space_size;
    b.m_texture = tex >= 0 ? &g_texs[tex] : nullptr;
    for (int i = 0; i < 64; i++) b.m_controls[i] = controls[i];
    b.m_normals = nullptr; b.m_intensities = nullptr;
    M_init_rendering(type);
    M_set_camera(0, 0, 0, 0, 0, 0, log_focus);
    if (as_object) {
        M_bicubic_object o;
        o.m_type = M_POLYGON_OBJECT; o.m_no_patches = 1; o.m_patches = &bp; o.m_centres = nullptr;
        M_init_bicubic_object(&o);
        M_shade_bicubic_object(&o, x, y, z, alp, bet, gam, nl, lp.data());
        M_render_bicubic_object(&o, x, y, z, alp, bet, gam);
        free(o.m_centres);
    } else {
        M_init_bicubic(&b);
        M_shade_bicubic(&b, x, y, z, alp, bet, gam, nl, lp.data());
        M_render_bicubic(&b, x, y, z, alp, bet, gam);
    }
    for (int i = 0; i < size * size * 4; i++) normals[i] = b.m_normals[i];
    for (int i = 0; i < size * size * 3; i++) intensities[i] = b.m_intensities[i];
    free(b.m_normals); free(b.m_intensities);
}

// host-only use (no GPU context): screen size and full-screen window for the clippers and T_perspective
void lte_screen_size(int w, int h) { HW_init_screen_size(w, h); C_init_clipping(0, 0, w - 1, h - 1); }
int lte_set_gate(int no_edges, int *indices, int *vertices, int log_focus)
{
    M_gate g;
    g.m_volume = nullptr; g.m_no_edges = no_edges; g.m_vertices = indices;
    M_set_camera(0, 0, 0, 0, 0, 0, log_focus);
    return M_set_gate(&g, vertices);
}
void lte_get_bounds(int *b) { C_get_bounds(b, b + 1, b + 2, b + 3); }
int lte_polygon_clipping(int axis, int *from, int *to, int dim, int len)
{
    return axis == 0 ? C_polygon_x_clipping(from, to, dim, len) : C_polygon_z_clipping(from, to, dim, len);
}
void lte_perspective(int *from, int *to, int dim, int len, int log_focus) { T_perspective(from, to, dim, len, log_focus); }
void lte_g_dot(int *v, int colour, int r, int g, int b) { G_dot(v, colour, r, g, b); }
void lte_g_pixel(int *v, int colour) { G_pixel(v, colour); }
void lte_g_line(int *v1, int *v2, int colour, int r, int g, int b) { G_line(v1, v2, colour, r, g, b); }

static void fill(M_polygon *p, M_texture *texs, const int *polys, const int *aux, int *pv, int i)
{
    p->m_type = aux ? (aux[i * 4 + 3] & 0xff) : M_PLANNAR;
    p->m_colour = polys[i * 4];
    p->m_red = aux ? aux[i * 4] : 0; p->m_green = aux ? aux[i * 4 + 1] : 0; p->m_blue = aux ? aux[i * 4 + 2] : 0;
    p->m_log_texture_space_size = aux ? (aux[i * 4 + 3] >> 8) : 0;
    p->m_texture = polys[i * 4 + 1] >= 0 ? &texs[polys[i * 4 + 1]] : nullptr;
    p->m_no_edges = polys[i * 4 + 2];
    p->m_vertices = pv + (size_t)polys[i * 4 + 3] * 6;
    p->m_normals = nullptr;
}

void lte_set_texture(int handle, int *texels, int log_size)
{
    g_texs[handle].m_texture_bytes = texels;
    g_texs[handle].m_log_texture_size = log_size;
}

// M_set_camera + M_render_polygon_object on one object
void lte_render_object(int *verts, int nv, const int *polys, const int *aux, int *pv, int np, int type,
                       int cam_alp, int cam_bet, int cam_gam, int cam_x, int cam_y, int cam_z, int log_focus,
                       int x, int y, int z, int alp, int bet, int gam)
{
    std::vector<M_polygon> store(np);
    std::vector<M_polygon *> ptrs(np);
    for (int i = 0; i < np; i++) { fill(&store[i], g_texs, polys, aux, pv, i); ptrs[i] = &store[i]; }
    M_polygon_object obj;
    obj.m_type = M_POLYGON_OBJECT; obj.m_order = nullptr;
    obj.m_no_polygons = np; obj.m_polygons = ptrs.data();
    obj.m_no_vertices = nv; obj.m_vertices = verts;
    obj.m_no_normals = 0; obj.m_normals = nullptr;
    M_init_rendering(type);
    M_set_camera((unsigned char)cam_alp, (unsigned char)cam_bet, (unsigned char)cam_gam, cam_x, cam_y, cam_z, log_focus);
    M_render_polygon_object(&obj, x, y, z, alp, bet, gam);
}

// M_shade_polygon_object on one object; polyverts R,G,B and aux r,g,b are written back
void lte_shade_object(int *verts, int nv, int *normals, int nn, const int *polys, int *aux, int *pv, int *pn, int np,
                      int x, int y, int z, int alp, int bet, int gam, const int *lights, int nl)
{
    std::vector<M_polygon> store(np);
    std::vector<M_polygon *> ptrs(np);
    for (int i = 0; i < np; i++) {
        fill(&store[i], g_texs, polys, aux, pv, i);
        store[i].m_normals = pn + polys[i * 4 + 3];
        ptrs[i] = &store[i];
    }
    std::vector<M_light> ls(nl);
    std::vector<M_light *> lp(nl);
    for (int j = 0; j < nl; j++) {
        ls[j].m_type = lights[j * 8];
        for (int k = 0; k < 4; k++) ls[j].m_parameter[k] = lights[j * 8 + 1 + k];
        ls[j].m_red = lights[j * 8 + 5]; ls[j].m_green = lights[j * 8 + 6]; ls[j].m_blue = lights[j * 8 + 7];
        lp[j] = &ls[j];
    }
    M_polygon_object obj;
    obj.m_type = M_POLYGON_OBJECT; obj.m_order = nullptr;
    obj.m_no_polygons = np; obj.m_polygons = ptrs.data();
    obj.m_no_vertices = nv; obj.m_vertices = verts;
    obj.m_no_normals = nn; obj.m_normals = normals;
    M_shade_polygon_object(&obj, x, y, z, alp, bet, gam, nl, lp.data());
    for (int i = 0; i < np; i++) { aux[i * 4] = store[i].m_red; aux[i * 4 + 1] = store[i].m_green; aux[i * 4 + 2] = store[i].m_blue; }
}

// M_init_polygon_object on one object; types[np] = m_type.  normals[1024][4] and
// polynormals[records] (entries of polygons without an m_normals array are left alone)
// are filled; returns m_no_normals.
int lte_init_object(int *verts, int nv, const int *polys, const int *types, int *pv, int np, int *normals, int *pn)
{
    std::vector<M_polygon> store(np);
    std::vector<M_polygon *> ptrs(np);
    for (int i = 0; i < np; i++) {
        fill(&store[i], g_texs, polys, nullptr, pv, i);
        store[i].m_type = types[i];
        store[i].m_normals = nullptr;
        ptrs[i] = &store[i];
    }
    M_polygon_object obj;
    obj.m_type = M_POLYGON_OBJECT; obj.m_order = nullptr;
    obj.m_no_polygons = np; obj.m_polygons = ptrs.data();
    obj.m_no_vertices = nv; obj.m_vertices = verts;
    obj.m_no_normals = 0; obj.m_normals = nullptr;
    M_init_polygon_object(&obj);
    for (int i = 0; i < obj.m_no_normals * 4; i++) normals[i] = obj.m_normals[i];
    for (int i = 0; i < np; i++)
        if (store[i].m_normals) {
            for (int k = 0; k <= store[i].m_no_edges; k++) pn[polys[i * 4 + 3] + k] = store[i].m_normals[k];
            free(store[i].m_normals);
        }
    free(obj.m_normals);
    return obj.m_no_normals;
}

// the transform, then one M_render_polygon call per polygon (as EngPoly.cpp:512-523 does)
void lte_render_polys_one_by_one(int *verts, int nv, const int *polys, const int *aux, int *pv, int np, int type,
                                 int log_focus, int x, int y, int z, int alp, int bet, int gam)
{
    std::vector<int> tmp((size_t)nv * 4);
    M_init_rendering(type);
    M_set_camera(0, 0, 0, 0, 0, 0, log_focus);
    T_set_self_rotation((unsigned char)alp, (unsigned char)bet, (unsigned char)gam);
    T_cancatinate_self_world(x + M_camera_x, y + M_camera_y, z + M_camera_z);
    T_cancatinated_rotation(verts, tmp.data(), nv);
    for (int i = 0; i < np; i++) {
        M_polygon p;
        fill(&p, g_texs, polys, aux, pv, i);
        M_render_polygon(&p, tmp.data(), nullptr, nullptr);
    }
}

void lte_g_polygon(int kind, int *tuples, int length, int colour, int r, int g, int b, int tex)
{
    if (kind == M_FLAT) G_flat_polygon(tuples, length, colour, r, g, b);
    else if (kind == M_SHADED) G_shaded_polygon(tuples, length, colour);
    else G_lin_textured_polygon(tuples, length, g_texs[tex].m_texture_bytes, g_texs[tex].m_log_texture_size);
}

void lte_g_prp_polygon(int *tuples, int length, int *O, int *u, int *v, int tex, int log_scale, int log_focus)
{
    M_camera_log_focus = log_focus;
    G_prp_textured_polygon(tuples, length, O, u, v, g_texs[tex].m_texture_bytes, g_texs[tex].m_log_texture_size, log_scale);
}

void lte_track_clipper_stores(int on) { HW_track_clipper_stores(on); }

}  // extern "C"

// Timing of a frame drawn as a reference-style program draws it: `frames` times G_clear and then
// M_render_polygon_object for every object of `per_object` polygons in turn (per_object >= np: the whole
// soup as one object), HW_flush at the end.  The M_polygon / M_polygon_object structures are built once,
// as such a program would hold them.  The soup's vertices lie three a triangle in polygon order;
// `pv_rel` holds the vertex records with indices counted from their object's first vertex.
// Returns milliseconds per frame (host clock around the whole loop, the final flush included); *calls_ms: of those, inside the calls.
extern "C" double lte_time_objects(int *verts, int nv, const int *polys, const int *aux, int *pv_rel, int np, int per_object,
                                   int type, int log_focus, int frames, double *calls_ms)
{
    std::vector<M_polygon> store(np);
    std::vector<M_polygon *> ptrs(np);
    for (int i = 0; i < np; i++) { fill(&store[i], g_texs, polys, aux, pv_rel, i); ptrs[i] = &store[i]; }
    if (per_object > np) per_object = np;
    std::vector<M_polygon_object> objs;
    for (int a = 0; a < np; a += per_object) {
        const int n = a + per_object <= np ? per_object : np - a;
        M_polygon_object o;
        o.m_type = M_POLYGON_OBJECT; o.m_order = nullptr;
        o.m_no_polygons = n; o.m_polygons = ptrs.data() + a;
        o.m_no_vertices = per_object >= np ? nv : 3 * n; o.m_vertices = verts + (per_object >= np ? 0 : (size_t)3 * a * 4);
        o.m_no_normals = 0; o.m_normals = nullptr;
        objs.push_back(o);
    }
    M_init_rendering(type);
    M_set_camera(0, 0, 0, 0, 0, 0, log_focus);
    struct timespec t0, t1;
    // one untimed frame first (allocations, code loading)
    G_clear(0, 255, 255, 255);
    for (M_polygon_object &o : objs) M_render_polygon_object(&o, 0, 0, 0, 0, 0, 0);
    HW_flush();
    clock_gettime(CLOCK_MONOTONIC, &t0);
    double in_calls = 0;
    for (int f = 0; f < frames; f++) {
        G_clear(0, 255, 255, 255);
        struct timespec c0, c1;
        clock_gettime(CLOCK_MONOTONIC, &c0);
        for (M_polygon_object &o : objs) M_render_polygon_object(&o, 0, 0, 0, 0, 0, 0);
        clock_gettime(CLOCK_MONOTONIC, &c1);
        in_calls += (c1.tv_sec - c0.tv_sec) * 1e3 + (c1.tv_nsec - c0.tv_nsec) * 1e-6;
    }
    HW_flush();
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (calls_ms) *calls_ms = in_calls / (frames > 0 ? frames : 1);     // inside the calls (the gather); the rest is the submit of the table and the GPU
    return ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6) / (frames > 0 ? frames : 1);
}
