/* Checker-side driver linked with the UNMODIFIED reference sources (compiled where
 * they lie under /root/reference by oracle/build_ref.sh) into
 * oracle/_ref/libltref_<W>x<H>_<tables>.so.
 *
 * Test infrastructure only: it exposes the reference's own functions
 * (Engine.h / Graphics.h / Trans.h / Clipper.h / Colour.h of the reference)
 * through a flat extern "C" surface so that Python tests can run the real
 * reference on the same inputs as the oracle restatement and the HIP path.
 * It supplies HW_error and HW_set_int, which the reference defines in its Win32
 * shell (LightTrack.cpp:56-69, 82-87).
 *
 * Flat scene layout shared with oracle/lt_oracle.c and include/lt_gpu.h:
 *   verts     int32[nv][4]        object or view space X Y Z (4th unused)
 *   polys     int32[np][4]        {colour, texture handle or -1, no_edges, first}
 *   polyaux   int32[np][4]        {m_red, m_green, m_blue, m_type | log_space<<8} (may be NULL)
 *   polyverts int32[*][6]         M_polygon.m_vertices records [Idx R G B Tx Ty];
 *                                 polygon i owns records first .. first+no_edges
 */
#include "LightTrack.h"
#include "Colour.h"
#include "Clipper.h"
#include "Graphics.h"
#include "Trans.h"
#include "Engine.h"
#include "data.h"
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include <limits.h>
#include <vector>

/* ---- platform functions the reference expects from its shell ---- */
void HW_error(char *s,...)
{
 va_list lst;
 va_start(lst,s);
 vfprintf(stderr,s,lst);
 va_end(lst);
 fprintf(stderr,"\n[ltref] HW_error: terminating (the reference exits here)\n");
 exit(3);
}

void HW_set_int(int *d,long l,int v)
{
 long i;
 for(i=0;i<l;i++) *d++=v;
}

/* ---- reference globals we read or set ---- */
extern int *G_z_buffer;
extern int G_miny,G_maxy;
extern int G_x_start[],G_x_end[];
extern HW_32_bit *G_start[],*G_end[];
extern HW_32_bit T_cx1,T_cx2,T_cx3,T_cy1,T_cy2,T_cy3,T_cz1,T_cz2,T_cz3;
extern HW_32_bit T_tx,T_ty,T_tz;
extern HW_32_bit T_sin[],T_cos[];
extern int M_rendering_type;
extern int C_2D_clipping,C_3D_clipping;
void GI_scan(int *edges,int dimension,int skip);
void GI_boarder_array_init(void);

#define LTREF_MAX_TEXTURES 64
static struct M_texture g_textures[LTREF_MAX_TEXTURES];
static std::vector<int> g_texture_store[LTREF_MAX_TEXTURES];
static int g_inited=0;

extern "C" {

/* the scene-description loader (data.cpp:676-781): the exported variable's memory */
void *ltref_d_data(const char *name) { return D_data((char*)name); }

int ltref_width(void)  { return HW_SCREEN_X_SIZE; }
int ltref_height(void) { return HW_SCREEN_Y_SIZE; }

void ltref_init(void)
{
 if(g_inited) return;
 T_init_math();
 CL_init_colour();
 G_init_graphics();
 C_init_clipping(C_X_CLIPPING_MIN,C_Y_CLIPPING_MIN,C_X_CLIPPING_MAX,C_Y_CLIPPING_MAX);
 g_inited=1;
}

void ltref_trig_tables(int *sin_out,int *cos_out)
{
 for(int i=0;i<256;i++){ sin_out[i]=T_sin[i]; cos_out[i]=T_cos[i]; }
}

int *ltref_cbuf(void) { return (int*)G_c_buffer; }
int *ltref_zbuf(void) { return G_z_buffer; }

void ltref_clear(int colour,int r,int g,int b) { G_clear(colour,r,g,b); }

int ltref_cl_light(int colour,int r,int g,int b) { return CL_light(colour,r,g,b); }
int ltref_cl_colour(int r,int g,int b) { return CL_colour(r,g,b); }

void ltref_set_texture(int handle,const int *texels,int log_size)
{
 size_t n=(size_t)1<<(2*log_size);
 g_texture_store[handle].assign(texels,texels+n);
 g_textures[handle].m_texture_bytes=g_texture_store[handle].data();
 g_textures[handle].m_log_texture_size=log_size;
}

/* camera + object placement -> the 12 coefficients of the concatenated transform */
void ltref_make_matrix(int cam_alp,int cam_bet,int cam_gam,int cam_x,int cam_y,int cam_z,
                       int log_focus,
                       int alp,int bet,int gam,int x,int y,int z,int *m12)
{
 M_set_camera((unsigned char)cam_alp,(unsigned char)cam_bet,(unsigned char)cam_gam,
              cam_x,cam_y,cam_z,log_focus);
 T_set_self_rotation((unsigned char)alp,(unsigned char)bet,(unsigned char)gam);
 T_cancatinate_self_world(x+M_camera_x,y+M_camera_y,z+M_camera_z);
 m12[0]=T_cx1; m12[1]=T_cy1; m12[2]=T_cz1; m12[3]=T_tx;
 m12[4]=T_cx2; m12[5]=T_cy2; m12[6]=T_cz2; m12[7]=T_ty;
 m12[8]=T_cx3; m12[9]=T_cy3; m12[10]=T_cz3; m12[11]=T_tz;
}

void ltref_set_matrix(const int *m12)
{
 T_cx1=m12[0]; T_cy1=m12[1]; T_cz1=m12[2]; T_tx=m12[3];
 T_cx2=m12[4]; T_cy2=m12[5]; T_cz2=m12[6]; T_ty=m12[7];
 T_cx3=m12[8]; T_cy3=m12[9]; T_cz3=m12[10]; T_tz=m12[11];
}

void ltref_transform(int *from,int *to,int n) { T_cancatinated_rotation(from,to,n); }

static void fill_polygon(struct M_polygon *p,const int *polys,const int *polyaux,
                         int *polyverts,int i)
{
 p->m_type=polyaux?(polyaux[i*4+3]&0xff):M_PLANNAR;
 p->m_colour=polys[i*4+0];
 p->m_red=polyaux?polyaux[i*4+0]:0;
 p->m_green=polyaux?polyaux[i*4+1]:0;
 p->m_blue=polyaux?polyaux[i*4+2]:0;
 p->m_log_texture_space_size=polyaux?(polyaux[i*4+3]>>8):0;
 p->m_texture=(polys[i*4+1]>=0)?&g_textures[polys[i*4+1]]:NULL;
 p->m_no_edges=polys[i*4+2];
 p->m_vertices=polyverts+(size_t)polys[i*4+3]*M_LNG_POLYGON_VERTEX;
 p->m_normals=NULL;
}

/* M_render_polygon over already transformed vertices, in array order (what
 * M_render_polygon_object does after its transform, EngPoly.cpp:518-523). */
void ltref_render_polys(int *viewverts,const int *polys,const int *polyaux,
                        int *polyverts,int np,int render_type,int log_focus,
                        int force_linear)
{
 struct M_polygon p;
 M_init_rendering(render_type);
 M_camera_log_focus=log_focus;
 M_force_linear_tmapping=force_linear;
 for(int i=0;i<np;i++)
 {
  fill_polygon(&p,polys,polyaux,polyverts,i);
  M_render_polygon(&p,viewverts,NULL,NULL);
 }
}

/* The real M_render_polygon_object (EngPoly.cpp:499-525); nv must stay < 1024. */
void ltref_render_object(int *verts,int nv,const int *polys,const int *polyaux,
                         int *polyverts,int np,int render_type,
                         int x,int y,int z,int alp,int bet,int gam)
{
 std::vector<struct M_polygon> store(np);
 std::vector<struct M_polygon*> ptrs(np);
 struct M_polygon_object obj;
 for(int i=0;i<np;i++){ fill_polygon(&store[i],polys,polyaux,polyverts,i); ptrs[i]=&store[i]; }
 obj.m_type=M_POLYGON_OBJECT; obj.m_order=NULL;
 obj.m_no_polygons=np; obj.m_polygons=ptrs.data();
 obj.m_no_vertices=nv; obj.m_vertices=verts;
 obj.m_no_normals=0; obj.m_normals=NULL;
 M_init_rendering(render_type);
 M_render_polygon_object(&obj,x,y,z,alp,bet,gam);
}

/* G_*_polygon level: kind 2 flat, 4 shaded, 8 linear textured (M_FLAT/M_SHADED/M_TEXTURED) */
void ltref_g_polygon(int kind,int *tuples,int length,int colour,int r,int g,int b,int tex)
{
 if(kind==M_FLAT) G_flat_polygon(tuples,length,colour,r,g,b);
 else if(kind==M_SHADED) G_shaded_polygon(tuples,length,colour);
 else G_lin_textured_polygon(tuples,length,g_textures[tex].m_texture_bytes,
                             g_textures[tex].m_log_texture_size);
}

void ltref_g_prp_polygon(int *tuples,int length,int *O,int *u,int *v,int tex,
                         int log_scale,int log_focus)
{
 M_camera_log_focus=log_focus;
 G_prp_textured_polygon(tuples,length,O,u,v,g_textures[tex].m_texture_bytes,
                        g_textures[tex].m_log_texture_size,log_scale);
}

/* ---- per-stage known-answer entry points ---- */
int ltref_volume_clipping(int *mn,int *mx) { return C_volume_clipping(mn,mx); }
int ltref_polygon_z_clipping(int *from,int *to,int dim,int len)
{ return C_polygon_z_clipping(from,to,dim,len); }
int ltref_polygon_x_clipping(int *from,int *to,int dim,int len)
{ return C_polygon_x_clipping(from,to,dim,len); }
void ltref_perspective(int *from,int *to,int dim,int len,int log_focus)
{ T_perspective(from,to,dim,len,log_focus); }
int ltref_normal_z_negative(int *a,int *b,int *c) { return T_normal_z_negative(a,b,c); }

/* line clippers: returns accept flag; out1/out2 get the clipped tuples; *first_moved
 * gets the "first vertex was replaced" flag (C_2D_clipping / C_3D_clipping). */
int ltref_line_clipping(int axis,int *v1,int *v2,int dim,int *out1,int *out2,int *first_moved)
{
 int *a=v1,*b=v2,r;
 if(axis==0)      { r=C_line_x_clipping(&a,&b,dim); *first_moved=C_2D_clipping; }
 else if(axis==1) { r=C_line_y_clipping(&a,&b,dim); *first_moved=C_2D_clipping; }
 else             { r=C_line_z_clipping(&a,&b,dim); *first_moved=C_3D_clipping; }
 if(r){ HW_copy_int(a,out1,dim); HW_copy_int(b,out2,dim); }
 return r;
}

/* x clip + edge scan of one screen polygon; returns miny/maxy and the row tables
 * for rows [0,H): xs, xe and start/end values of the dim-2 interpolated values. */
void ltref_scan_polygon(int *tuples,int length,int dim,int *miny,int *maxy,
                        int *xs,int *xe,int *vstart,int *vend)
{
 int new_edges[G_MAX_POLYGON_VERTICES*G_LNG_TEXTURED];
 int n,i,y,j;
 GI_boarder_array_init();
 n=C_polygon_x_clipping(tuples,new_edges,dim,length);
 for(i=0;i<n;i++) GI_scan(&new_edges[i*dim],dim,dim);
 *miny=G_miny; *maxy=G_maxy;
 for(y=0;y<HW_SCREEN_Y_SIZE;y++)
 {
  xs[y]=G_x_start[y]; xe[y]=G_x_end[y];
  for(j=0;j<dim-2;j++)
  {
   vstart[y*6+j]=(xs[y]!=INT_MAX)?G_start[j][y]:0;
   vend[y*6+j]=(xe[y]!=INT_MIN)?G_end[j][y]:0;
  }
 }
}

/* C_init_clipping (Clipp2D.cpp:36-42): the 2-D clip window of everything drawn next */
void ltref_init_clipping(int minx,int miny,int maxx,int maxy)
{
 C_init_clipping(minx,miny,maxx,maxy);
}

/* The real M_set_gate (EngBase.cpp:567-629) on view-space vertices, starting from the window
 * (minx, miny, maxx, maxy); bounds[4] receives the window it leaves (C_get_bounds).  The
 * full-screen window is restored afterwards. */
int ltref_set_gate(int no_edges,int *indices,int *vertices,int log_focus,const int *start,int *bounds)
{
 struct M_gate g;
 int rc;
 g.m_volume=NULL; g.m_no_edges=no_edges; g.m_vertices=indices;
 M_set_camera(0,0,0,0,0,0,log_focus);
 C_init_clipping(start[0],start[1],start[2],start[3]);
 rc=M_set_gate(&g,vertices);
 C_get_bounds(bounds,bounds+1,bounds+2,bounds+3);
 C_init_clipping(C_X_CLIPPING_MIN,C_Y_CLIPPING_MIN,C_X_CLIPPING_MAX,C_Y_CLIPPING_MAX);
 return rc;
}

/* The real M_init_bicubic, M_shade_bicubic and M_render_bicubic (EngBcub.cpp:204-403) on one
 * patch: controls[16][4]; normals[size*size][4] and intensities[size*size][3] are copied out;
 * the patch is drawn with the current frame, rendering type `type`, camera at the origin.
 * lights[nl][8] as in ltref_shade_object. */
void ltref_bicubic(const int *controls,int log_render_size,int colour,int log_texture_space_size,int tex,
                   int x,int y,int z,int alp,int bet,int gam,const int *lights,int nl,
                   int type,int log_focus,int force_linear,int *normals,int *intensities)
{
 struct M_bicubic b;
 std::vector<struct M_light> lstore(nl);
 std::vector<struct M_light*> lptrs(nl);
 int size=(1<<log_render_size)+1,old=M_force_linear_tmapping;
 for(int j=0;j<nl;j++)
 {
  lstore[j].m_type=lights[j*8];
  for(int k=0;k<4;k++) lstore[j].m_parameter[k]=lights[j*8+1+k];
  lstore[j].m_red=lights[j*8+5]; lstore[j].m_green=lights[j*8+6]; lstore[j].m_blue=lights[j*8+7];
  lptrs[j]=&lstore[j];
 }
 b.m_log_render_size=log_render_size;
 b.m_colour=colour;
 b.m_log_texture_space_size=log_texture_space_size;
 b.m_texture=(tex>=0)?&g_textures[tex]:NULL;
 for(int i=0;i<16*T_LNG_VECTOR;i++) b.m_controls[i]=controls[i];
 b.m_normals=NULL; b.m_intensities=NULL;
 M_init_bicubic(&b);
 for(int i=0;i<size*size*T_LNG_VECTOR;i++) normals[i]=b.m_normals[i];
 M_shade_bicubic(&b,x,y,z,alp,bet,gam,nl,lptrs.data());
 for(int i=0;i<size*size*3;i++) intensities[i]=b.m_intensities[i];
 M_init_rendering(type);
 M_set_camera(0,0,0,0,0,0,log_focus);
 M_force_linear_tmapping=force_linear;
 M_render_bicubic(&b,x,y,z,alp,bet,gam);
 M_force_linear_tmapping=old;
 free(b.m_normals); free(b.m_intensities);
}

/* The real G_dot and G_pixel (GraphicsBase.cpp:117-153) */
void ltref_g_dot(int *v,int colour,int r,int g,int b){ G_dot(v,colour,r,g,b); }
void ltref_g_pixel(int *v,int colour){ G_pixel(v,colour); }

/* The real G_line (GraphicsBase.cpp:165-253); v1, v2 = projected {X, Y, Z} */
void ltref_g_line(int *v1,int *v2,int colour,int r,int g,int b)
{
 G_line(v1,v2,colour,r,g,b);
}

/* The real M_init_polygon_object (EngPoly.cpp:355-459) without _PAINTER_: normals only.
 * types[np] = m_type.  normals[1024][4] receives the object's list (three words per
 * normal are defined), polynormals[*] (laid out like polyverts) the indices of polygons
 * that got an m_normals array.  Returns m_no_normals. */
int ltref_init_object(int *verts,int nv,const int *polys,const int *types,int *polyverts,int np,
                      int *normals,int *polynormals)
{
 std::vector<struct M_polygon> store(np);
 std::vector<struct M_polygon*> ptrs(np);
 struct M_polygon_object obj;
 for(int i=0;i<np;i++)
 {
  fill_polygon(&store[i],polys,NULL,polyverts,i);
  store[i].m_type=types[i];
  ptrs[i]=&store[i];
 }
 obj.m_type=M_POLYGON_OBJECT; obj.m_order=NULL;
 obj.m_no_polygons=np; obj.m_polygons=ptrs.data();
 obj.m_no_vertices=nv; obj.m_vertices=verts;
 obj.m_no_normals=0; obj.m_normals=NULL;
 M_init_polygon_object(&obj);
 for(int i=0;i<obj.m_no_normals;i++)
  for(int k=0;k<3;k++) normals[i*4+k]=obj.m_normals[i*T_LNG_VECTOR+k];
 for(int i=0;i<np;i++)
  if(store[i].m_normals)
  {
   for(int k=0;k<=store[i].m_no_edges;k++) polynormals[polys[i*4+3]+k]=store[i].m_normals[k];
   free(store[i].m_normals);
  }
 free(obj.m_normals);
 return obj.m_no_normals;
}

} /* extern "C" */

/* ---- shading pre-pass (EngPoly.cpp:464-493) ---- */
extern HW_32_bit T_sx1,T_sx2,T_sx3,T_sy1,T_sy2,T_sy3,T_sz1,T_sz2,T_sz3;

extern "C" {

/* T_set_self_rotation with the tables of this build -> s9 = sx1 sy1 sz1 sx2 .. sz3 */
void ltref_self_matrix(int alp,int bet,int gam,int *s9)
{
 T_set_self_rotation((unsigned char)alp,(unsigned char)bet,(unsigned char)gam);
 s9[0]=T_sx1; s9[1]=T_sy1; s9[2]=T_sz1;
 s9[3]=T_sx2; s9[4]=T_sy2; s9[5]=T_sz2;
 s9[6]=T_sx3; s9[7]=T_sy3; s9[8]=T_sz3;
}

/* The real M_shade_polygon_object; nv and nn must stay within M_MAX_OBJECT_VERTICES.
 * lights[nl][8] = {type, px,py,pz,pw, red,green,blue}; polyverts R,G,B and polyaux
 * m_red,m_green,m_blue are overwritten. */
void ltref_shade_object(int *verts,int nv,int *normals,int nn,const int *polys,int *polyaux,
                        int *polyverts,int *polynormals,int np,
                        int x,int y,int z,int alp,int bet,int gam,const int *lights,int nl)
{
 std::vector<struct M_polygon> store(np);
 std::vector<struct M_polygon*> ptrs(np);
 std::vector<struct M_light> lstore(nl);
 std::vector<struct M_light*> lptrs(nl);
 struct M_polygon_object obj;
 for(int i=0;i<np;i++)
 {
  fill_polygon(&store[i],polys,polyaux,polyverts,i);
  store[i].m_normals=polynormals+polys[i*4+3];
  ptrs[i]=&store[i];
 }
 for(int j=0;j<nl;j++)
 {
  lstore[j].m_type=lights[j*8];
  for(int k=0;k<4;k++) lstore[j].m_parameter[k]=lights[j*8+1+k];
  lstore[j].m_red=lights[j*8+5]; lstore[j].m_green=lights[j*8+6]; lstore[j].m_blue=lights[j*8+7];
  lptrs[j]=&lstore[j];
 }
 obj.m_type=M_POLYGON_OBJECT; obj.m_order=NULL;
 obj.m_no_polygons=np; obj.m_polygons=ptrs.data();
 obj.m_no_vertices=nv; obj.m_vertices=verts;
 obj.m_no_normals=nn; obj.m_normals=normals;
 M_shade_polygon_object(&obj,x,y,z,alp,bet,gam,nl,lptrs.data());
 for(int i=0;i<np;i++)
 {
  polyaux[i*4+0]=store[i].m_red; polyaux[i*4+1]=store[i].m_green; polyaux[i*4+2]=store[i].m_blue;
 }
}

} /* extern "C" */
