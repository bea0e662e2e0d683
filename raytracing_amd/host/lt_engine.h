// Host-side mirror of the reference's polygon-path API, backed by the MI355X library
// through the C-ABI of include/lt_gpu.h.
//
// A program written against the reference's Engine.h / Graphics.h / Trans.h / Clipper.h /
// Colour.h (C++ linkage, global state, immediate-mode drawing) includes this header
// instead and links liblt_engine.so + libltgpu.so.  Names, argument meaning and the error
// behaviour (HW_error terminates the process, LightTrack.cpp:56-69) are the reference's.
// What differs, and why:
//  * the screen size is a run-time value: call HW_init_screen_size(w, h) before
//    G_init_graphics (the reference fixes it at compile time, LightTrack.h:44-45);
//  * drawing is deferred: draw calls are queued and executed on the GPU when the frame is
//    observed -- HW_flush() (the place of the reference's HW_blit, LightTrack.cpp:284-298)
//    -- or when G_clear is called.  After HW_flush, G_c_buffer and G_z_buffer hold the
//    frame in host memory exactly as the reference leaves them;
//  * the clip window (C_init_clipping, C_set_bounds) must be a non-empty part of the screen:
//    the reference would write outside its buffers with any other window.
#pragma once
#include <stdint.h>

typedef int HW_32_bit;
typedef HW_32_bit HW_pixel;                 // _32BPP_, LightTrack.h:38-42

// ---- platform (the reference's LightTrack.h)
void HW_init_screen_size(int width, int height);
int HW_screen_x_size(void);
int HW_screen_y_size(void);
void HW_error(char *s, ...);                // prints and terminates (weak: a harness may replace it)
void HW_flush(void);                        // run queued draws, bring the frame to G_c_buffer / G_z_buffer
void HW_select_device(int device);
// G_line clips with dimension 2, so the depth of a clipped end point is a leftover of earlier
// clipper calls (Clipp2D.cpp:99-104).  on: every polygon drawn also makes the reference's
// clipper calls on the host and G_line clips there, so that a clipped line drawn after filled
// polygons gets the depth the reference reads (host work per polygon).  off (default): a
// clipped end point takes depth 0, the value in a program that draws only wireframes.
void HW_track_clipper_stores(int on);          // before G_init_graphics; default 0

// ---- Colour.h:140-148
void CL_init_colour(void);
HW_pixel CL_colour(int red_light, int green_light, int blue_light);
HW_pixel CL_light(HW_pixel colour, int red_light, int green_light, int blue_light);

// ---- Graphics.h:37-113
#define G_MAX_POLYGON_VERTICES 16
#define G_Z_MAPPING_SWITCH 1024
#define G_P 16
#define G_LNG_FLAT 3
#define G_LNG_SHADED 6
#define G_LNG_TEXTURED 8
extern HW_pixel *G_c_buffer;
extern int *G_z_buffer;
extern long G_c_buffer_size;
void G_init_graphics(void);
void G_clear(HW_pixel colour, int red, int green, int blue);
void G_pixel(int *vertex, HW_pixel colour);
void G_dot(int *vertex, HW_pixel colour, int red, int green, int blue);
void G_line(int *vertex1, int *vertex2, HW_pixel colour, int red, int green, int blue);
void G_flat_polygon(int *edges, int length, HW_pixel colour, int red, int green, int blue);
void G_shaded_polygon(int *edges, int length, HW_pixel colour);
void G_lin_textured_polygon(int *edges, int length, HW_pixel *texture, int log_texture_size);
void G_prp_textured_polygon(int *edges, int length, int *O, int *u, int *v, HW_pixel *texture, int log_texture_size,
                            int log_texture_scale);       // Graphics.h:108-113; reads M_camera_log_focus

// ---- Trans.h:27-72 (the parts on the polygon path)
#define T_LNG_VECTOR 4
#define T_P 15
void T_init_math(void);
void T_set_self_rotation(unsigned char alp, unsigned char bet, unsigned char gam);
void T_set_world_rotation(unsigned char alp, unsigned char bet, unsigned char gam);
void T_cancatinate_self_world(int addx, int addy, int addz);
void T_cancatinated_rotation(int *from, int *to, int length);
void T_perspective(int *from, int *to, int dimension, int length, int log_focus);    // Trans.h:62
int T_normal_z_negative(int *from1, int *from2, int *from3);                         // Trans.h:42

// ---- Clipper.h:26-28
void C_init_clipping(int minx, int miny, int maxx, int maxy);
void C_get_bounds(int *minx, int *miny, int *maxx, int *maxy);
int C_set_bounds(int minx, int miny, int maxx, int maxy);
// Clipper.h:29-35, on the host (the GPU path has its own copies): binary-search clippers that
// redirect the caller's pointers to static stores, and the polygon clippers built on them
extern int C_2D_clipping, C_3D_clipping;
int C_line_x_clipping(int **vertex1, int **vertex2, int dimension);
int C_line_y_clipping(int **vertex1, int **vertex2, int dimension);
int C_polygon_x_clipping(int *from, int *to, int dimension, int length);
int C_volume_clipping(int *min, int *max);
int C_line_z_clipping(int **vertex1, int **vertex2, int dimension);
int C_polygon_z_clipping(int *from, int *to, int dimension, int length);

// ---- Engine.h:80-90, 148-152, 209-238, 435-460
#define M_WIRE 0x01
#define M_FLAT 0x02
#define M_SHADED 0x04
#define M_TEXTURED 0x08
#define M_PLANNAR 0x01
#define M_CURVED 0x02
#define M_QUAD_XY 0x04
#define M_QUAD_MINUS_XY 0x08
#define M_LNG_POLYGON_VERTEX 6
#define M_MAX_OBJECT_VERTICES 1024
#define M_POLYGON_OBJECT 0x1

extern int M_force_linear_tmapping;
extern int M_camera_x, M_camera_y, M_camera_z, M_camera_log_focus;

struct M_texture {
    HW_pixel *m_texture_bytes;
    int m_log_texture_size;
};

struct M_polygon {
    int m_type;
    HW_pixel m_colour;
    int m_red, m_green, m_blue;
    int m_log_texture_space_size;
    struct M_texture *m_texture;
    int m_no_edges;
    int *m_vertices;            // [Idx R G B Tx Ty] x (m_no_edges + 1)
    int *m_normals;
};

struct M_polygon_object_order;
struct M_polygon_object {
    int m_type;
    struct M_polygon_object_order *m_order;     // must be NULL (no BSP tree with a depth buffer)
    int m_no_polygons;
    struct M_polygon **m_polygons;
    int m_no_vertices;
    int *m_vertices;
    int m_no_normals;
    int *m_normals;
};

// Engine.h:112-130 (_RGB_)
#define M_AMBIENT 0x1
#define M_POINT   0x2
#define M_DIRECT  0x4
struct M_light {
    int m_type;
    int m_parameter[4];
    int m_red, m_green, m_blue;
};

// Engine.h:342-350, EngBase.cpp:567-629: narrows the clip window to a portal polygon's extent
struct M_volume_object;
struct M_gate {
    struct M_volume_object *m_volume;
    int m_no_edges;
    int *m_vertices;            // closed list of vertex indices
};
int M_set_gate(struct M_gate *gate, int *vertices);

// EngPoly.cpp:355-459 (no BSP ordering with a depth buffer): gives every M_PLANNAR / M_CURVED
// polygon its m_normals indices and the object its list of normals; m_normals must be NULL
// on entry, as in the reference.
void M_init_polygon_object(struct M_polygon_object *object);

// EngPoly.cpp:464-493: every polygon's vertex records receive their R,G,B and the polygon
// its flat m_red/m_green/m_blue, computed on the GPU and written back before the call returns.
void M_shade_polygon_object(struct M_polygon_object *object, int x, int y, int z, int alp, int bet, int gam,
                            int no_lights, struct M_light **lights);

// Engine.h:78, 275-300, 490-516 and EngBcub.cpp: bicubic Bezier patches, tessellated into squares
#define M_PB 8
#define M_MAX_RENDER_SIZE 33
#define M_MAX_PATCHES 128
struct M_bicubic {
    int m_log_render_size;
    HW_pixel m_colour;
    int m_log_texture_space_size;
    struct M_texture *m_texture;
    int m_controls[16 * T_LNG_VECTOR];
    int *m_normals;
    int *m_intensities;
};
struct M_bicubic_object {
    int m_type;
    int m_no_patches;
    struct M_bicubic **m_patches;
    int *m_centres;
};
void M_init_bicubic(struct M_bicubic *bicubic);
void M_shade_bicubic(struct M_bicubic *bicubic, int x, int y, int z, int alp, int bet, int gam, int no_lights,
                     struct M_light **lights);
void M_render_bicubic(struct M_bicubic *bicubic, int x, int y, int z, int alp, int bet, int gam);
void M_init_bicubic_object(struct M_bicubic_object *object);
void M_shade_bicubic_object(struct M_bicubic_object *object, int x, int y, int z, int alp, int bet, int gam, int no_lights,
                            struct M_light **lights);
void M_render_bicubic_object(struct M_bicubic_object *object, int x, int y, int z, int alp, int bet, int gam);

void M_init_rendering(int type);
void M_set_camera(unsigned char alp, unsigned char bet, unsigned char gam, int x, int y, int z, int log_focus);
void M_render_polygon(struct M_polygon *polygon, int *vertices, int *colours, int *textures);
void M_render_polygon_object(struct M_polygon_object *object, int x, int y, int z, int alp, int bet, int gam);
