/* CPU restatement of the reference's polygon rasterization path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP library: only the
 * tests, the smoke check and the benchmark's cpu_baseline leg may load it; the product
 * library never links or calls it.
 *
 * Parity unpinned, in the strict sense: the reference ships no tests, golden vectors or
 * sample output for this path, and its sources compile here only beside a stand-in for its
 * Win32 platform header (oracle/shadow/LightTrack.h: the image has no <windows.h>).  What
 * this file IS checked against: the reference's own .cpp files, compiled unchanged where
 * they lie by oracle/build_ref.sh (tests/test_oracle_vs_ref.py fuzzes every function
 * here against that build), and the vectors under tests/golden/ that the same build
 * produced.
 *
 * It is a plain sequential restatement, in the same order of operations as the
 * reference, with the screen size as a run-time parameter (the reference fixes it at
 * compile time, LightTrack.h:44-45) and with 32-bit wrap-around made explicit.  All
 * arithmetic is integer; ">>" is an arithmetic shift, "/" truncates toward zero, and
 * add, subtract, multiply and "<<" wrap modulo 2^32 (the reference build this is pinned
 * against is compiled -fwrapv on an LP64 target, so its "long" is 64 bits wide).
 *
 * Build configuration restated: _RGB_ _32BPP_ _FIXED_ _Z_BUFFER_ _MONO_
 * (LightTrack.dsp:46).  Build: gcc -O2 -fwrapv -shared -fPIC (see __graft_entry__.build).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <math.h>

#define LTO_MAX_DIM       8      /* C_MAX_DIMENSIONS, Clipper.h:17 */
#define LTO_MAX_TUPLES   16      /* G_MAX_POLYGON_VERTICES, Graphics.h:37 */
#define LTO_WORK_TUPLES  64      /* working room; more than 16 sets the overflow flag */
#define LTO_GP           16      /* G_P, Graphics.h:39 */
#define LTO_TP           15      /* T_P, Trans.h:31 */
#define LTO_Z_MIN        10      /* C_Z_CLIPPING_MIN, Clipper.h:19 */
#define LTO_Z_CONST (-0x7ffff)   /* T_ONE_OVER_Z_CONST, Trans.h:30 */
#define LTO_Z_SWITCH   1024      /* G_Z_MAPPING_SWITCH, Graphics.h:38 */
#define LTO_LINEAR       32      /* G_LINEAR, GraphicsPoly.cpp:40 */
#define LTO_M_WIRE        1      /* Engine.h:80-83 */
#define LTO_M_FLAT        2
#define LTO_M_SHADED      4
#define LTO_M_TEXTURED    8
#define LTO_M_QUAD_XY     4      /* Engine.h:193-195 */
#define LTO_M_QUAD     0x0c
#define LTO_MAX_TEXTURES 64

typedef int32_t i32;
typedef uint32_t u32;
typedef int64_t i64;

static inline i32 wadd(i32 a,i32 b){ return (i32)((u32)a+(u32)b); }
static inline i32 wsub(i32 a,i32 b){ return (i32)((u32)a-(u32)b); }
static inline i32 wmul(i32 a,i32 b){ return (i32)((u32)a*(u32)b); }
static inline i32 wshl(i32 a,int s){ return (i32)((u32)a<<s); }
/* truncating division that cannot trap on INT_MIN/-1 (wraps like the hardware the
 * reference ran on would not: that case never occurs with divisors >= 1) */
static inline i32 tdiv(i32 a,i32 b){ return (i32)((i64)a/(i64)b); }

typedef struct lto_ctx {
    int w,h;
    i32 *cbuf,*zbuf;
    i32 *xs,*xe;                 /* G_x_start / G_x_end, GraphicsPoly.cpp:44-45 */
    i32 *vs[6],*ve[6];           /* G_0..5_start / _end, GraphicsPoly.cpp:46-57 */
    int miny,maxy;
    int log_focus;               /* M_camera_log_focus */
    i32 cx0,cy0,cx1,cy1;         /* C_x_clipping_min .. C_y_clipping_max, Clipp2D.cpp:23-26 */
    int fast_reset;              /* 1: reset only the rows a polygon touched */
    int touched_lo,touched_hi;
    const i32 *tex[LTO_MAX_TEXTURES];
    int tex_log[LTO_MAX_TEXTURES];
    uint64_t fragments,passes,polys_in,polys_drawn,overflow,unsupported;
} lto_ctx;

/* ------------------------------------------------------------------ colour */

/* CL_light, Colour.cpp:93-102 with the table of Colour.cpp:44-56 evaluated
 * arithmetically: CL_div[c][l] = (c*l)/255, light clamped from above only
 * (Colour.h:135). */
i32 lto_cl_light(i32 colour,i32 r,i32 g,i32 b)
{
    i32 cr=(colour>>16)&0xff, cg=(colour>>8)&0xff, cb=colour&0xff;
    if(r>255) r=255;
    if(g>255) g=255;
    if(b>255) b=255;
    return (((cr*r)/255)<<16) | (((cg*g)/255)<<8) | ((cb*b)/255);
}

/* CL_colour, Colour.cpp:67-76 */
i32 lto_cl_colour(i32 r,i32 g,i32 b)
{
    if(r>255) r=255;
    if(g>255) g=255;
    if(b>255) b=255;
    return (r<<16)|(g<<8)|b;
}

/* ------------------------------------------------------------------ context */

lto_ctx *lto_create(int w,int h)
{
    lto_ctx *c=(lto_ctx*)calloc(1,sizeof(lto_ctx));
    int j;
    if(!c) return NULL;
    c->w=w; c->h=h;
    c->cbuf=(i32*)malloc(sizeof(i32)*(size_t)w*h);     /* G_init_graphics, GraphicsBase.cpp:70-89 */
    c->zbuf=(i32*)malloc(sizeof(i32)*(size_t)w*h);
    c->xs=(i32*)malloc(sizeof(i32)*h);
    c->xe=(i32*)malloc(sizeof(i32)*h);
    for(j=0;j<6;j++){ c->vs[j]=(i32*)calloc(h,sizeof(i32)); c->ve[j]=(i32*)calloc(h,sizeof(i32)); }
    for(j=0;j<h;j++){ c->xs[j]=INT_MAX; c->xe[j]=INT_MIN; }
    c->touched_lo=0; c->touched_hi=-1;
    c->cx0=0; c->cy0=0; c->cx1=w-1; c->cy1=h-1;       /* Clipper.h:19-23 */
    return c;
}

void lto_destroy(lto_ctx *c)
{
    int j;
    if(!c) return;
    free(c->cbuf); free(c->zbuf); free(c->xs); free(c->xe);
    for(j=0;j<6;j++){ free(c->vs[j]); free(c->ve[j]); }
    free(c);
}

i32 *lto_cbuf(lto_ctx *c){ return c->cbuf; }
i32 *lto_zbuf(lto_ctx *c){ return c->zbuf; }
void lto_set_fast_reset(lto_ctx *c,int on){ c->fast_reset=on; }
/* C_init_clipping, Clipp2D.cpp:36-42; the window must lie inside the screen */
void lto_set_clip_window(lto_ctx *c,i32 minx,i32 miny,i32 maxx,i32 maxy)
{ c->cx0=minx; c->cy0=miny; c->cx1=maxx; c->cy1=maxy; }
void lto_set_texture(lto_ctx *c,int handle,const i32 *texels,int log_size)
{ c->tex[handle]=texels; c->tex_log[handle]=log_size; }

void lto_counters(lto_ctx *c,uint64_t *out)
{
    out[0]=c->fragments; out[1]=c->passes; out[2]=c->polys_in;
    out[3]=c->polys_drawn; out[4]=c->overflow; out[5]=c->unsupported;
}
void lto_reset_counters(lto_ctx *c)
{ c->fragments=c->passes=c->polys_in=c->polys_drawn=c->overflow=c->unsupported=0; }

/* G_clear, GraphicsBase.cpp:99-111 */
void lto_clear(lto_ctx *c,i32 colour,i32 r,i32 g,i32 b)
{
    size_t n=(size_t)c->w*c->h,i;
    i32 v=lto_cl_light(colour,r,g,b);
    for(i=0;i<n;i++){ c->cbuf[i]=v; c->zbuf[i]=INT_MAX; }
}

/* ------------------------------------------------------------------ transforms */

/* T_init_math: variant 0 follows Trans.cpp:62-77 (scale, then truncate), variant 1
 * follows TransBs.cpp:63-78 (truncate, then scale). */
void lto_trig_tables(int variant,i32 *s,i32 *co)
{
    int i;
    for(i=0;i<256;i++)
    {
        if(variant==0){ s[i]=(i32)(sin(i/40.7436642)*(1<<LTO_TP)); co[i]=(i32)(cos(i/40.7436642)*(1<<LTO_TP)); }
        else { s[i]=((int)sin(i/40.7436642))*(1<<LTO_TP); co[i]=((int)cos(i/40.7436642))*(1<<LTO_TP); }
    }
}

#define FX(a,b) (wmul((a),(b))>>LTO_TP)

/* T_set_self_rotation, TransBs.cpp:144-183 -> s[9] = sx1 sy1 sz1 sx2 sy2 sz2 sx3 sy3 sz3 */
static void self_rotation(const i32 *S,const i32 *C,int alp,int bet,int gam,i32 *s)
{
    i32 ca=C[alp&255],sa=S[alp&255],cb=C[bet&255],sb=S[bet&255],cg=C[gam&255],sg=S[gam&255];
    s[0]=wsub(FX(ca,cg),FX(sa,FX(sb,sg)));
    s[1]=wadd(FX(sa,cg),FX(ca,FX(sb,sg)));
    s[2]=FX(cb,sg);
    s[3]=-FX(sa,cb);
    s[4]=FX(ca,cb);
    s[5]=-sb;
    s[6]=wsub(-FX(ca,sg),FX(sa,FX(sb,cg)));
    s[7]=wsub(FX(ca,FX(sb,cg)),FX(sa,sg));
    s[8]=FX(cb,cg);
}

/* T_set_world_rotation, TransBs.cpp:239-278 -> w[9] = wx1 wy1 wz1 wx2 wy2 wz2 wx3 wy3 wz3 */
static void world_rotation(const i32 *S,const i32 *C,int alp,int bet,int gam,i32 *w)
{
    i32 ca=C[alp&255],sa=S[alp&255],cb=C[bet&255],sb=S[bet&255],cg=C[gam&255],sg=S[gam&255];
    w[0]=wadd(FX(sg,FX(sb,sa)),FX(cg,ca));
    w[1]=FX(cb,sa);
    w[2]=wsub(FX(sg,ca),FX(cg,FX(sb,sa)));
    w[3]=wsub(FX(sg,FX(sb,ca)),FX(cg,sa));
    w[4]=FX(cb,ca);
    w[5]=wsub(-FX(cg,FX(sb,ca)),FX(sg,sa));
    w[6]=-FX(sg,cb);
    w[7]=sb;
    w[8]=FX(cg,cb);
}

/* M_set_camera (EngBase.cpp:201-213) + T_set_self_rotation +
 * T_cancatinate_self_world (TransBs.cpp:325-361) as M_render_polygon_object
 * calls them (EngPoly.cpp:510-511).
 * m12 rows: {cx1 cy1 cz1 tx, cx2 cy2 cz2 ty, cx3 cy3 cz3 tz}. */
void lto_make_matrix(int variant,int cam_alp,int cam_bet,int cam_gam,
                     i32 cam_x,i32 cam_y,i32 cam_z,
                     int alp,int bet,int gam,i32 x,i32 y,i32 z,i32 *m)
{
    i32 S[256],C[256],s[9],w[9],ax,ay,az;
    int r;
    lto_trig_tables(variant,S,C);
    world_rotation(S,C,(unsigned char)(256-cam_alp),(unsigned char)(256-cam_bet),
                   (unsigned char)(256-cam_gam),w);
    self_rotation(S,C,(unsigned char)alp,(unsigned char)bet,(unsigned char)gam,s);
    ax=wadd(x,-cam_x); ay=wadd(y,-cam_y); az=wadd(z,-cam_z);
    for(r=0;r<3;r++)
    {   /* column r of W is (w[3r], w[3r+1], w[3r+2]) = (wx,wy,wz)_{r+1} */
        i32 wx=w[3*r],wy=w[3*r+1],wz=w[3*r+2];
        m[4*r+0]=wadd(wadd(FX(s[0],wx),FX(s[3],wy)),FX(s[6],wz));   /* cx: sx1 sx2 sx3 */
        m[4*r+1]=wadd(wadd(FX(s[1],wx),FX(s[4],wy)),FX(s[7],wz));   /* cy: sy1 sy2 sy3 */
        m[4*r+2]=wadd(wadd(FX(s[2],wx),FX(s[5],wy)),FX(s[8],wz));   /* cz: sz1 sz2 sz3 */
        m[4*r+3]=wadd(wadd(FX(ax,wx),FX(ay,wy)),FX(az,wz));
    }
}

/* T_cancatinated_rotation, TransBs.cpp:373-393; the 4th element is left alone. */
void lto_transform(const i32 *m,const i32 *from,i32 *to,int n)
{
    int i,r;
    for(i=0;i<n;i++,from+=4,to+=4)
    {
        i32 x=from[0],y=from[1],z=from[2];
        for(r=0;r<3;r++)
            to[r]=wadd(wadd(wadd(FX(m[4*r],x),FX(m[4*r+1],y)),FX(m[4*r+2],z)),m[4*r+3]);
    }
}

/* T_perspective, TransBs.cpp:416-434 */
void lto_perspective(const lto_ctx *c,const i32 *from,i32 *to,int dim,int len,int log_focus)
{
    int i,j;
    for(i=0;i<len;i++,from+=dim,to+=dim)
    {
        i64 z=from[2];
        to[0]=wadd((i32)(((i64)from[0]*((i64)1<<log_focus))/z),c->w/2);
        to[1]=wadd((i32)(((i64)from[1]*((i64)1<<log_focus))/z),c->h/2);
        to[2]=(i32)((i64)LTO_Z_CONST/z);
        for(j=3;j<dim;j++) to[j]=from[j];
    }
}

/* T_normal_z_negative, TransLn.cpp:218-223 */
int lto_normal_z_negative(const i32 *a,const i32 *b,const i32 *c)
{
    return ((i64)wsub(b[0],a[0])*(i64)wsub(c[1],a[1])-
            (i64)wsub(b[1],a[1])*(i64)wsub(c[0],a[0]))<0;
}

/* ------------------------------------------------------------------ clipping */

/* C_volume_clipping, Clipp3D.cpp:46-58 */
int lto_volume_clipping(const i32 *mn,const i32 *mx)
{
    if((mx[2]<mn[0])||(mx[2]<mn[1])||(mx[2]<-mx[0])||(mx[2]<-mx[1])||(mx[2]<=LTO_Z_MIN)) return 0;
    return (mn[2]<LTO_Z_MIN)?-1:1;
}

static void midpoint(i32 *m,const i32 *l,const i32 *r,int dim)
{ int i; for(i=0;i<dim;i++) m[i]=wadd(l[i],r[i])>>1; }

/* One routine for C_line_x_clipping / C_line_y_clipping (Clipp2D.cpp:94-228; both
 * sides, axis 0 or 1) and C_line_z_clipping (Clipp3D.cpp:70-110; near side only,
 * axis 2).  Binary search on every dimension until the clipped coordinate equals the
 * bound.  *first_moved restates C_2D_clipping / C_3D_clipping. */
static int clip_line(const i32 *v1,const i32 *v2,int dim,int axis,i32 bmin,i32 bmax,
                     int both,i32 *o1,i32 *o2,int *first_moved)
{
    i32 lo[LTO_MAX_DIM],hi[LTO_MAX_DIM],l[LTO_MAX_DIM],r[LTO_MAX_DIM],m[LTO_MAX_DIM];
    int swap,whereto;
    *first_moved=0;
    if(v1[axis]<v2[axis]){ swap=0; memcpy(lo,v1,dim*4); memcpy(hi,v2,dim*4); }
    else                 { swap=1; memcpy(lo,v2,dim*4); memcpy(hi,v1,dim*4); }
    if(both){ if((lo[axis]>=bmax)||(hi[axis]<=bmin)) return 0; }
    else    { if(hi[axis]<=bmin) return 0; }

    if(lo[axis]<=bmin)
    {
        memcpy(m,lo,dim*4); memcpy(r,hi,dim*4);
        whereto=1;
        while(m[axis]!=bmin)
        {
            if(whereto) memcpy(l,m,dim*4); else memcpy(r,m,dim*4);
            midpoint(m,l,r,dim);
            whereto=m[axis]<bmin;
        }
        memcpy(lo,m,dim*4);
        *first_moved=swap^1;
    }
    if(both&&(hi[axis]>=bmax))
    {
        memcpy(l,lo,dim*4); memcpy(m,hi,dim*4);
        whereto=0;
        while(m[axis]!=bmax)
        {
            if(whereto) memcpy(l,m,dim*4); else memcpy(r,m,dim*4);
            midpoint(m,l,r,dim);
            whereto=m[axis]<bmax;
        }
        memcpy(hi,m,dim*4);
        *first_moved|=swap&1;
    }
    if(swap){ memcpy(o1,hi,dim*4); memcpy(o2,lo,dim*4); }
    else    { memcpy(o1,lo,dim*4); memcpy(o2,hi,dim*4); }
    return 1;
}

int lto_line_clipping(const lto_ctx *c,int axis,const i32 *v1,const i32 *v2,int dim,
                      i32 *o1,i32 *o2,int *first_moved)
{
    if(axis==0) return clip_line(v1,v2,dim,0,c->cx0,c->cx1,1,o1,o2,first_moved);
    if(axis==1) return clip_line(v1,v2,dim,1,c->cy0,c->cy1,1,o1,o2,first_moved);
    return clip_line(v1,v2,dim,2,LTO_Z_MIN,0,0,o1,o2,first_moved);
}

/* C_polygon_z_clipping (Clipp3D.cpp:129-159) and C_polygon_x_clipping
 * (Clipp2D.cpp:247-275): `to` must hold LTO_WORK_TUPLES tuples; returns the new
 * edge count (may exceed what the reference's arrays could hold: caller checks). */
static int clip_polygon(const lto_ctx *c,int axis,const i32 *from,i32 *to,int dim,int len)
{
    int i,n=0,moved;
    i32 a[LTO_MAX_DIM],b[LTO_MAX_DIM];
    for(i=0;i<len;i++)
    {
        if(lto_line_clipping(c,axis,from+i*dim,from+(i+1)*dim,dim,a,b,&moved))
        {
            if(n+3>LTO_WORK_TUPLES) return LTO_WORK_TUPLES;     /* out of working room */
            if(moved){ memcpy(to+n*dim,a,dim*4); n++; memcpy(to+n*dim,b,dim*4); n++; }
            else     { memcpy(to+n*dim,b,dim*4); n++; }
        }
    }
    if(n>0) memcpy(to+n*dim,to,dim*4);
    return n;
}

int lto_polygon_z_clipping(const lto_ctx *c,const i32 *from,i32 *to,int dim,int len)
{ return clip_polygon(c,2,from,to,dim,len); }
int lto_polygon_x_clipping(const lto_ctx *c,const i32 *from,i32 *to,int dim,int len)
{ return clip_polygon(c,0,from,to,dim,len); }

/* ------------------------------------------------------------------ edge scan */

/* GI_boarder_array_init, GraphicsPoly.cpp:209-216 */
static void border_init(lto_ctx *c)
{
    int y;
    c->miny=INT_MAX; c->maxy=INT_MIN;
    if(c->fast_reset)
    {
        for(y=c->touched_lo;y<=c->touched_hi;y++){ c->xs[y]=INT_MAX; c->xe[y]=INT_MIN; }
        c->touched_lo=c->h; c->touched_hi=-1;
    }
    else
        for(y=0;y<c->h;y++){ c->xs[y]=INT_MAX; c->xe[y]=INT_MIN; }
}

/* GI_scan, GraphicsPoly.cpp:124-200.  dim = number of values per tuple that are
 * interpolated (incl. X Y), skip = tuple stride. */
static void scan_edge(lto_ctx *c,const i32 *edge,int dim,int skip)
{
    i32 a[LTO_MAX_DIM],b[LTO_MAX_DIM],cur[6],inc[6];
    int moved,i,j,nv=dim-2;
    i32 x,y,dx,dy,long_d,short_d,d,add_dh,add_dl,ixh,iyh,ixl,iyl;

    if(!clip_line(edge,edge+skip,dim,1,c->cy0,c->cy1,1,a,b,&moved)) return;

    x=a[0]; y=a[1]; dx=b[0]; dy=b[1];
    if(y<c->miny) c->miny=y;
    if(y>c->maxy) c->maxy=y;
    if(dy<c->miny) c->miny=dy;
    if(dy>c->maxy) c->maxy=dy;
    if(c->fast_reset)
    {
        i32 lo=y<dy?y:dy,hi=y<dy?dy:y;
        if(lo<c->touched_lo) c->touched_lo=lo;
        if(hi>c->touched_hi) c->touched_hi=hi;
    }
    dx-=x; dy-=y;
    if(dx<0){ dx=-dx; ixh=-1; ixl=-1; } else { ixh=1; ixl=1; }
    if(dy<0){ dy=-dy; iyh=-1; iyl=-1; } else { iyh=1; iyl=1; }
    if(dx>dy){ long_d=dx; short_d=dy; iyl=0; } else { long_d=dy; short_d=dx; ixl=0; }
    d=2*short_d-long_d; add_dl=2*short_d; add_dh=2*(short_d-long_d);

    for(j=0;j<6;j++) inc[j]=0;                      /* unset in the reference when long_d==0; never used then */
    cur[0]=wshl(a[2],LTO_GP);
    if(long_d>0) inc[0]=tdiv(wshl(wsub(b[2],a[2]),LTO_GP),long_d);
    for(j=1;j<nv;j++)
    {
        cur[j]=a[2+j];
        if(long_d>0) inc[j]=tdiv(wsub(b[2+j],a[2+j]),long_d);
    }
    for(i=0;i<=long_d;i++)
    {
        if(x<c->xs[y]){ c->xs[y]=x; for(j=0;j<nv;j++) c->vs[j][y]=cur[j]; }
        if(c->xe[y]<x){ c->xe[y]=x; for(j=0;j<nv;j++) c->ve[j][y]=cur[j]; }
        if(d>=0){ x+=ixh; y+=iyh; d+=add_dh; } else { x+=ixl; y+=iyl; d+=add_dl; }
        for(j=0;j<nv;j++) cur[j]=wadd(cur[j],inc[j]);
    }
}

/* x clip + scan of all edges; returns 0 if the clipped polygon would not fit the
 * reference's 16-tuple arrays (the reference would overrun its stack there). */
static int scan_polygon(lto_ctx *c,const i32 *tuples,int length,int dim,int scan_dim)
{
    i32 clipped[LTO_WORK_TUPLES*LTO_MAX_DIM+LTO_MAX_DIM];
    int n,i;
    border_init(c);
    n=clip_polygon(c,0,tuples,clipped,dim,length);
    if(n+1>LTO_MAX_TUPLES){ c->overflow++; c->miny=INT_MAX; c->maxy=INT_MIN; return 0; }
    for(i=0;i<n;i++) scan_edge(c,clipped+i*dim,scan_dim,dim);
    return 1;
}

void lto_scan_polygon(lto_ctx *c,const i32 *tuples,int length,int dim,int *miny,int *maxy,
                      i32 *xs,i32 *xe,i32 *vstart,i32 *vend)
{
    int y,j;
    scan_polygon(c,tuples,length,dim,dim);
    *miny=c->miny; *maxy=c->maxy;
    for(y=0;y<c->h;y++)
    {
        xs[y]=c->xs[y]; xe[y]=c->xe[y];
        for(j=0;j<dim-2;j++)
        {
            vstart[y*6+j]=(xs[y]!=INT_MAX)?c->vs[j][y]:0;
            vend[y*6+j]=(xe[y]!=INT_MIN)?c->ve[j][y]:0;
        }
    }
}

/* ------------------------------------------------------------------ fills */

/* G_flat_polygon, GraphicsPoly.cpp:230-291 */
void lto_flat_polygon(lto_ctx *c,const i32 *tuples,int length,i32 colour,i32 r,i32 g,i32 b)
{
    i32 hw=lto_cl_light(colour,r,g,b);
    int y;
    if(!scan_polygon(c,tuples,length,3,3)) return;
    if(!(c->miny<c->maxy)) return;
    for(y=c->miny;y<c->maxy;y++)
    {
        i32 beg=c->xs[y],end=c->xe[y],span=wsub(end,beg),cz,iz=0,x;
        size_t pos=(size_t)y*c->w;
        if(beg==INT_MAX) continue;                   /* row never touched: the reference's loop body does not run */
        cz=c->vs[0][y];
        if(span!=0) iz=tdiv(wsub(c->ve[0][y],cz),span);
        for(x=beg;x<=end;x++)
        {
            c->fragments++;
            if(c->zbuf[pos+x]>(cz>>LTO_GP)){ c->zbuf[pos+x]=cz>>LTO_GP; c->cbuf[pos+x]=hw; c->passes++; }
            cz=wadd(cz,iz);
        }
    }
}

/* G_shaded_polygon, GraphicsPoly.cpp:300-384 */
void lto_shaded_polygon(lto_ctx *c,const i32 *tuples,int length,i32 colour)
{
    int y;
    if(!scan_polygon(c,tuples,length,6,6)) return;
    if(!(c->miny<c->maxy)) return;
    for(y=c->miny;y<c->maxy;y++)
    {
        i32 beg=c->xs[y],end=c->xe[y],span=wsub(end,beg),x;
        i32 cz,cr,cg,cb,iz,ir,ig,ib;
        size_t pos=(size_t)y*c->w;
        if(beg==INT_MAX) continue;                   /* row never touched: the reference's loop body does not run */
        if(span==0) span=1;
        cr=c->vs[1][y]; ir=tdiv(wsub(c->ve[1][y],cr),span);
        cg=c->vs[2][y]; ig=tdiv(wsub(c->ve[2][y],cg),span);
        cb=c->vs[3][y]; ib=tdiv(wsub(c->ve[3][y],cb),span);
        cz=c->vs[0][y]; iz=tdiv(wsub(c->ve[0][y],cz),span);
        for(x=beg;x<=end;x++)
        {
            c->fragments++;
            if(c->zbuf[pos+x]>(cz>>LTO_GP))
            {
                c->zbuf[pos+x]=cz>>LTO_GP;
                c->cbuf[pos+x]=lto_cl_light(colour,cr>>LTO_GP,cg>>LTO_GP,cb>>LTO_GP);
                c->passes++;
            }
            cr=wadd(cr,ir); cg=wadd(cg,ig); cb=wadd(cb,ib); cz=wadd(cz,iz);
        }
    }
}

/* G_lin_textured_polygon, GraphicsPoly.cpp:398-502.  The light accumulators advance
 * only when the depth test passes (GraphicsPoly.cpp:485-493). */
void lto_lin_textured_polygon(lto_ctx *c,const i32 *tuples,int length,const i32 *texture,int log_size)
{
    i32 mask=wsub(wshl(1,log_size+LTO_GP),1);
    int y;
    if(!scan_polygon(c,tuples,length,8,8)) return;
    if(!(c->miny<c->maxy)) return;
    for(y=c->miny;y<c->maxy;y++)
    {
        i32 beg=c->xs[y],end=c->xe[y],span=wsub(end,beg),x;
        i32 cz,cr,cg,cb,tx,ty,iz,ir,ig,ib,itx,ity;
        size_t pos=(size_t)y*c->w;
        if(beg==INT_MAX) continue;
        if(span==0) span=1;
        tx=c->vs[4][y]; itx=tdiv(wsub(c->ve[4][y],tx),span);
        ty=c->vs[5][y]; ity=tdiv(wsub(c->ve[5][y],ty),span);
        cr=c->vs[1][y]; ir=tdiv(wsub(c->ve[1][y],cr),span);
        cg=c->vs[2][y]; ig=tdiv(wsub(c->ve[2][y],cg),span);
        cb=c->vs[3][y]; ib=tdiv(wsub(c->ve[3][y],cb),span);
        cz=c->vs[0][y]; iz=tdiv(wsub(c->ve[0][y],cz),span);
        for(x=beg;x<=end;x++)
        {
            tx&=mask; ty&=mask;
            c->fragments++;
            if(c->zbuf[pos+x]>(cz>>LTO_GP))
            {
                c->zbuf[pos+x]=cz>>LTO_GP;
                c->cbuf[pos+x]=lto_cl_light(texture[((ty>>LTO_GP)<<log_size)+(tx>>LTO_GP)],
                                            cr>>LTO_GP,cg>>LTO_GP,cb>>LTO_GP);
                cr=wadd(cr,ir); cg=wadd(cg,ig); cb=wadd(cb,ib);
                c->passes++;
            }
            tx=wadd(tx,itx); ty=wadd(ty,ity); cz=wadd(cz,iz);
        }
    }
}

/* texture coordinate at perspective-space column x of one row,
 * GraphicsPoly.cpp:618-622 / 630-634 */
static void prp_point(i32 x,i32 xx,i32 yy,i32 zz,i32 ui,i32 vi,i32 zi,int lf,int log_scale,
                      i32 u0,i32 v0,i32 *tx,i32 *ty)
{
    i32 zzz=wadd(zz,wmul(x,zi)>>lf);
    if(zzz!=0)
    {
        *tx=wadd(wshl(tdiv(wshl(wadd(xx,wmul(x,ui)>>lf),log_scale),zzz),LTO_GP),u0);
        *ty=wadd(wshl(tdiv(wshl(wadd(yy,wmul(x,vi)>>lf),log_scale),zzz),LTO_GP),v0);
    }
    else { *tx=0; *ty=0; }
}

/* G_prp_textured_polygon, GraphicsPoly.cpp:520-684 */
void lto_prp_textured_polygon(lto_ctx *c,const i32 *tuples,int length,const i32 *O,
                              const i32 *u,const i32 *v,const i32 *texture,
                              int log_size,int log_scale)
{
    i32 mask=wsub(wshl(1,log_size+LTO_GP),1);
    i32 x0=O[0],y0=O[1],z0=O[2],u0=O[6],v0=O[7];
    i32 Vx=v[0],Vy=v[1],Vz=v[2],Ux=u[0],Uy=u[1],Uz=u[2];
    i32 ui,uj,uc,vi,vj,vc,zi,zj,zc;
    int lf=c->log_focus,row;
    if(!scan_polygon(c,tuples,length,8,6)) return;
    if(!(c->miny<c->maxy)) return;
    ui=wsub(wmul(Vz,y0),wmul(Vy,z0)); uj=wsub(wmul(Vx,z0),wmul(Vz,x0)); uc=wsub(wmul(Vy,x0),wmul(Vx,y0));
    vi=wsub(wmul(Uy,z0),wmul(Uz,y0)); vj=wsub(wmul(Uz,x0),wmul(Ux,z0)); vc=wsub(wmul(Ux,y0),wmul(Uy,x0));
    zi=wsub(wmul(Vy,Uz),wmul(Vz,Uy)); zj=wsub(wmul(Vz,Ux),wmul(Vx,Uz)); zc=wsub(wmul(Vx,Uy),wmul(Vy,Ux));
    for(row=c->miny;row<c->maxy;row++)
    {
        i32 beg=c->xs[row],end=c->xe[row],span=wsub(end,beg),x,y,px;
        i32 cz,cr,cg,cb,iz,ir,ig,ib,tx,ty,itx,ity,etx,ety,xx,yy,zz;
        size_t pos=(size_t)row*c->w;
        if(beg==INT_MAX) continue;
        if(span==0) span=1;
        px=beg;                                  /* screen column of the next pixel */
        beg=wsub(beg,c->w/2); x=beg; end=wsub(end,c->w/2); y=row-c->h/2;
        cr=c->vs[1][row]; ir=tdiv(wsub(c->ve[1][row],cr),span);
        cg=c->vs[2][row]; ig=tdiv(wsub(c->ve[2][row],cg),span);
        cb=c->vs[3][row]; ib=tdiv(wsub(c->ve[3][row],cb),span);
        cz=c->vs[0][row]; iz=tdiv(wsub(c->ve[0][row],cz),span);
        xx=wadd(wmul(y,uj)>>lf,uc); yy=wadd(wmul(y,vj)>>lf,vc); zz=wadd(wmul(y,zj)>>lf,zc);
        prp_point(x,xx,yy,zz,ui,vi,zi,lf,log_scale,u0,v0,&etx,&ety);
        for(;beg<=end;)
        {
            x+=LTO_LINEAR; if(x>end) x=end;
            tx=etx; ty=ety;
            prp_point(x,xx,yy,zz,ui,vi,zi,lf,log_scale,u0,v0,&etx,&ety);
            if((span=x-beg)!=0){ itx=tdiv(wsub(etx,tx),span); ity=tdiv(wsub(ety,ty),span); }
            else { itx=ity=0; }
            for(;beg<=x;beg++,px++)
            {
                tx&=mask; ty&=mask;
                c->fragments++;
                if(c->zbuf[pos+px]>(cz>>LTO_GP))
                {
                    c->zbuf[pos+px]=cz>>LTO_GP;
                    c->cbuf[pos+px]=lto_cl_light(texture[((ty>>LTO_GP)<<log_size)+(tx>>LTO_GP)],
                                                 cr>>LTO_GP,cg>>LTO_GP,cb>>LTO_GP);
                    cr=wadd(cr,ir); cg=wadd(cg,ig); cb=wadd(cb,ib);
                    c->passes++;
                }
                tx=wadd(tx,itx); ty=wadd(ty,ity); cz=wadd(cz,iz);
            }
        }
    }
}

/* kind: 2 flat, 4 shaded, 8 linear textured */
void lto_g_polygon(lto_ctx *c,int kind,const i32 *tuples,int length,i32 colour,
                   i32 r,i32 g,i32 b,int tex)
{
    if(kind==LTO_M_FLAT) lto_flat_polygon(c,tuples,length,colour,r,g,b);
    else if(kind==LTO_M_SHADED) lto_shaded_polygon(c,tuples,length,colour);
    else lto_lin_textured_polygon(c,tuples,length,c->tex[tex],c->tex_log[tex]);
}

void lto_g_prp_polygon(lto_ctx *c,const i32 *tuples,int length,const i32 *O,const i32 *u,
                       const i32 *v,int tex,int log_scale,int log_focus)
{
    c->log_focus=log_focus;
    lto_prp_textured_polygon(c,tuples,length,O,u,v,c->tex[tex],c->tex_log[tex],log_scale);
}

/* ------------------------------------------------------------------ engine */

/* T_linear_solve, TransBs.cpp:440-515 (fixed-point branch), n x n system, m right
 * hand sides.  Returns 0 where the reference would divide by zero. */
static int linear_solve(i32 ia[8][8],i32 ib[8][8],i32 ix[8][8],int n,int m,int log_src,int log_res)
{
    i32 a[8][8],b[8][8],x[8][8],max,tmp,pivot,sum;
    int i,j,k,num;
    for(i=0;i<n;i++)
    {
        for(j=0;j<n;j++) a[i][j]=ia[i][j]>>log_src;
        for(j=0;j<m;j++) b[i][j]=ib[i][j];
    }
    for(max=0,num=0,i=0;i<n-1;i++)
    {
        for(j=i;j<n;j++)
        {
            if(a[j][i]>=0) pivot=a[j][i]; else pivot=-a[j][i];
            if(pivot>max){ max=pivot; num=j; }
        }
        if(max!=0)
        {
            if(num!=i)
            {
                for(j=0;j<n;j++){ tmp=a[i][j]; a[i][j]=a[num][j]; a[num][j]=tmp; }
                for(j=0;j<m;j++){ tmp=b[i][j]; b[i][j]=b[num][j]; b[num][j]=tmp; }
            }
            for(j=i+1;j<n;j++)
            {
                if(a[i][i]==0) return 0;
                for(k=i+1;k<n;k++) a[j][k]=wsub(a[j][k],tdiv(wmul(a[i][k],a[j][i]),a[i][i]));
                for(k=0;k<m;k++)   b[j][k]=wsub(b[j][k],tdiv(wmul(b[i][k],a[j][i]),a[i][i]));
            }
        }
    }
    for(i=n-1;i>=0;i--)
    {
        for(k=0;k<m;k++)
        {
            for(sum=0,j=i+1;j<n;j++) sum=wadd(sum,wmul(a[i][j],x[j][k])>>log_res);
            if(a[i][i]==0) return 0;
            x[i][k]=tdiv(wshl(wsub(b[i][k],sum),log_res),a[i][i]);
            ix[i][k]=x[i][k];
        }
    }
    return 1;
}

/* MI_tmapping_vectors, EngBase.cpp:145-179; p1..p3 are M_LNG_TEXTURED tuples */
static int tmapping_vectors(int type,const i32 *p1,const i32 *p2,const i32 *p3,
                            int log_space,i32 *u,i32 *v)
{
    int j;
    if(type)
    {
        if(type&LTO_M_QUAD_XY){ for(j=0;j<3;j++){ u[j]=wsub(p2[j],p1[j]); v[j]=wsub(p3[j],p2[j]); } }
        else                  { for(j=0;j<3;j++){ u[j]=wsub(p1[j],p2[j]); v[j]=wsub(p2[j],p3[j]); } }
        return 1;
    }
    else
    {
        i32 a[8][8],b[8][8],x[8][8];
        memset(a,0,sizeof(a)); memset(b,0,sizeof(b)); memset(x,0,sizeof(x));
        for(j=0;j<2;j++){ a[0][j]=wsub(p2[6+j],p1[6+j]); a[1][j]=wsub(p3[6+j],p2[6+j]); }
        for(j=0;j<3;j++){ b[0][j]=wsub(p2[j],p1[j]); b[1][j]=wsub(p3[j],p2[j]); }
        if(!linear_solve(a,b,x,2,3,LTO_GP,log_space)) return 0;
        for(j=0;j<3;j++){ u[j]=x[0][j]; v[j]=x[1][j]; }
        return 1;
    }
}

/* CI_construct_tuples (EngBase.cpp:47-138, colours==NULL, textures==NULL) for one
 * polygon: rec = M_polygon.m_vertices records [Idx R G B Tx Ty]. */
static void construct_tuples(const i32 *rec,i32 *to,const i32 *verts,int dim,int len,i32 *mn,i32 *mx)
{
    int i,j;
    for(j=0;j<3;j++){ mn[j]=INT_MAX; mx[j]=INT_MIN; }
    for(i=0;i<len;i++,rec+=6,to+=dim)
    {
        const i32 *p=verts+(size_t)rec[0]*4;
        for(j=0;j<3;j++)
        {
            to[j]=p[j];
            if(to[j]>mx[j]) mx[j]=to[j];
            if(to[j]<mn[j]) mn[j]=to[j];
        }
        if(dim>=6){ to[3]=wshl(rec[1],LTO_GP); to[4]=wshl(rec[2],LTO_GP); to[5]=wshl(rec[3],LTO_GP); }
        if(dim==8){ to[6]=wshl(rec[4],LTO_GP); to[7]=wshl(rec[5],LTO_GP); }
    }
}

/* The two clips of G_line (GraphicsBase.cpp:191-193): C_line_x_clipping / C_line_y_clipping
 * with dimension 2, so only X and Y of an end point that a clipper moves are computed; the
 * pointer then refers to one of the clipper's static stores (Clipp2D.cpp:99-104,132,150)
 * and the depth read through it is whatever that store's third word holds.  In a program
 * that draws only wireframes those words are never written and stay 0 (static storage);
 * after filled polygons they hold leftovers of the polygon clipper's calls.  This
 * restatement, like the HIP path, uses 0. */
static int clip_line_xy(i32 *v1,i32 *v2,int axis,i32 bmin,i32 bmax)
{
    i32 a[2]={v1[0],v1[1]},b[2]={v2[0],v2[1]},o1[2],o2[2];
    int moved,swap=!(v1[axis]<v2[axis]);
    int m1=swap?(v1[axis]>=bmax):(v1[axis]<=bmin);
    int m2=swap?(v2[axis]<=bmin):(v2[axis]>=bmax);
    if(!clip_line(a,b,2,axis,bmin,bmax,1,o1,o2,&moved)) return 0;
    v1[0]=o1[0]; v1[1]=o1[1]; if(m1) v1[2]=0;
    v2[0]=o2[0]; v2[1]=o2[1]; if(m2) v2[2]=0;
    return 1;
}

/* G_line, GraphicsBase.cpp:165-253 (_RGB_, _Z_BUFFER_): depth is tested but not written. */
void lto_g_line(lto_ctx *c,const i32 *vertex1,const i32 *vertex2,i32 colour,i32 r,i32 g,i32 b)
{
    i32 v1[3]={vertex1[0],vertex1[1],vertex1[2]},v2[3]={vertex2[0],vertex2[1],vertex2[2]};
    i32 hw=lto_cl_light(colour,r,g,b);
    i32 dx,dy,long_d,short_d,d,add_dh,add_dl,cur_z,inc_z=0;
    int inc_xh,inc_yh,inc_xl,inc_yl,x,y,i;
    if(!clip_line_xy(v1,v2,0,c->cx0,c->cx1)) return;
    if(!clip_line_xy(v1,v2,1,c->cy0,c->cy1)) return;
    dx=v2[0]-v1[0]; dy=v2[1]-v1[1];
    if(dx<0){ dx=-dx; inc_xh=-1; inc_xl=-1; } else { inc_xh=1; inc_xl=1; }
    if(dy<0){ dy=-dy; inc_yh=-1; inc_yl=-1; } else { inc_yh=1; inc_yl=1; }
    if(dx>dy){ long_d=dx; short_d=dy; inc_yl=0; } else { long_d=dy; short_d=dx; inc_xl=0; }
    d=2*short_d-long_d; add_dl=2*short_d; add_dh=2*(short_d-long_d);
    x=v1[0]; y=v1[1];
    cur_z=v1[2];
    if(long_d!=0) inc_z=tdiv(wsub(v2[2],cur_z),long_d);
    for(i=0;i<=long_d;i++)
    {
        size_t pos=(size_t)y*c->w+x;
        c->fragments++;
        if(cur_z<c->zbuf[pos]) c->cbuf[pos]=hw;
        if(d>=0){ d+=add_dh; x+=inc_xh; y+=inc_yh; }
        else    { d+=add_dl; x+=inc_xl; y+=inc_yl; }
        cur_z=wadd(cur_z,inc_z);
    }
}

/* G_dot, GraphicsBase.cpp:133-153 (_RGB_, _Z_BUFFER_) */
void lto_g_dot(lto_ctx *c,const i32 *vertex,i32 colour,i32 r,i32 g,i32 b)
{
    if(vertex[0]>=0&&vertex[0]<c->w&&vertex[1]>=0&&vertex[1]<c->h)
    {
        size_t pos=(size_t)vertex[1]*c->w+vertex[0];
        c->fragments++;
        if(vertex[2]<c->zbuf[pos]) c->cbuf[pos]=lto_cl_light(colour,r,g,b);
    }
}

/* G_pixel, GraphicsBase.cpp:117-123; the caller keeps the point on the screen */
void lto_g_pixel(lto_ctx *c,const i32 *vertex,i32 colour)
{
    if(vertex[0]>=0&&vertex[0]<c->w&&vertex[1]>=0&&vertex[1]<c->h)
        c->cbuf[(size_t)vertex[1]*c->w+vertex[0]]=colour;
}

/* M_render_polygon, EngBase.cpp:379-552, for one polygon of the flat scene layout.
 * polys[i] = {colour, texture handle or -1, no_edges, first};
 * polyaux[i] = {m_red, m_green, m_blue, m_type | log_texture_space_size<<8}. */
static void render_polygon(lto_ctx *c,const i32 *viewverts,const i32 *poly,const i32 *aux,
                           const i32 *polyverts,int render_type,int force_linear)
{
    i32 original1[LTO_WORK_TUPLES*LTO_MAX_DIM+LTO_MAX_DIM];
    i32 original2[LTO_WORK_TUPLES*LTO_MAX_DIM+LTO_MAX_DIM];
    i32 persp[LTO_WORK_TUPLES*LTO_MAX_DIM+LTO_MAX_DIM];
    i32 mn[3],mx[3],u[4],v[4],*clipped;
    const i32 *rec=polyverts+(size_t)poly[3]*6;
    int number=poly[2],cnd,dim,kind=render_type;

    c->polys_in++;
    if(number+1>LTO_MAX_TUPLES){ c->overflow++; return; }
    if(kind==LTO_M_TEXTURED&&poly[1]<0) kind=LTO_M_SHADED;      /* EngBase.cpp:470,518-522 */
    if(kind==LTO_M_FLAT||kind==LTO_M_WIRE) dim=3; else if(kind==LTO_M_SHADED) dim=6; else if(kind==LTO_M_TEXTURED) dim=8;
    else { c->unsupported++; return; }

    construct_tuples(rec,original1,viewverts,dim,number+1,mn,mx);
    if((cnd=lto_volume_clipping(mn,mx))==0) return;
    if(cnd==-1)
    {
        number=clip_polygon(c,2,original1,original2,dim,number);
        if(number+1>LTO_MAX_TUPLES){ c->overflow++; return; }
        clipped=original2;
    }
    else clipped=original1;
    /* With fewer than two edges left the reference's back-face test reads tuples it
     * never wrote (EngBase.cpp:541-543); that cannot happen for a closed polygon that
     * passed the volume test, and is dropped here and in the HIP path. */
    if(number<2) return;
    lto_perspective(c,clipped,persp,dim,number+1,c->log_focus);
    if(!lto_normal_z_negative(persp,persp+dim,persp+2*dim)) return;
    c->polys_drawn++;
    if(kind==LTO_M_WIRE)                                         /* EngBase.cpp:416-428 */
    {
        int i;
        for(i=0;i<number;i++)
            lto_g_line(c,persp+i*3,persp+(i+1)*3,poly[0],aux?aux[0]:0,aux?aux[1]:0,aux?aux[2]:0);
    }
    else if(kind==LTO_M_FLAT)
        lto_flat_polygon(c,persp,number,poly[0],aux?aux[0]:0,aux?aux[1]:0,aux?aux[2]:0);
    else if(kind==LTO_M_SHADED)
        lto_shaded_polygon(c,persp,number,poly[0]);
    else
    {
        if(force_linear||(clipped[2]<LTO_Z_SWITCH))              /* EngBase.cpp:491-493 */
        {
            int type=aux?(aux[3]&0xff):0,log_space=aux?(aux[3]>>8):0;
            if(!tmapping_vectors(type&LTO_M_QUAD,original1,original1+8,original1+16,log_space,u,v))
            { c->unsupported++; return; }
            lto_prp_textured_polygon(c,persp,number,clipped,u,v,c->tex[poly[1]],c->tex_log[poly[1]],log_space);
        }
        else lto_lin_textured_polygon(c,persp,number,c->tex[poly[1]],c->tex_log[poly[1]]);
    }
}

/* The polygon loop of M_render_polygon_object (EngPoly.cpp:518-523) over vertices
 * that lto_transform has already moved into view space. */
void lto_render_polys(lto_ctx *c,const i32 *viewverts,const i32 *polys,const i32 *polyaux,
                      const i32 *polyverts,int np,int render_type,int log_focus,int force_linear)
{
    int i;
    c->log_focus=log_focus;
    for(i=0;i<np;i++)
        render_polygon(c,viewverts,polys+(size_t)i*4,polyaux?polyaux+(size_t)i*4:NULL,
                       polyverts,render_type,force_linear);
}

/* ------------------------------------------------------------------ shading pre-pass
 *
 * M_shade_polygon_object (EngPoly.cpp:464-493): vertices and normals are rotated by the
 * object's self rotation, vertices translated to world space, then every polygon's vertex
 * records receive R,G,B from the light list, and the polygon its flat illumination.
 * Lights: {type, px, py, pz, pw, red, green, blue} (struct M_light, Engine.h:116-130);
 * type 1 ambient, 2 point, 4 direct (M_AMBIENT / M_POINT / M_DIRECT, Engine.h:112-114);
 * any other type contributes nothing (the switch of M_shade_vertex has no default). */

#define LTO_LOG_NORMAL 7                    /* T_LOG_NORMAL_SIZE, Trans.h:29 */

/* T_set_self_rotation alone (TransBs.cpp:144-183): s[9] = sx1 sy1 sz1 sx2 ... sz3 */
void lto_self_matrix(int variant,int alp,int bet,int gam,i32 *s)
{
    i32 S[256],C[256];
    lto_trig_tables(variant,S,C);
    self_rotation(S,C,(unsigned char)alp,(unsigned char)bet,(unsigned char)gam,s);
}

/* T_self_rotation, TransBs.cpp:194-214 (products wrap in 32 bits, then >>T_P) */
static void self_rotate(const i32 *s,const i32 *from,i32 *to)
{
    i32 x=from[0],y=from[1],z=from[2];
    to[0]=wadd(wadd(FX(s[0],x),FX(s[1],y)),FX(s[2],z));
    to[1]=wadd(wadd(FX(s[3],x),FX(s[4],y)),FX(s[5],z));
    to[2]=wadd(wadd(FX(s[6],x),FX(s[7],y)),FX(s[8],z));
}

/* TI_sqrt, TransLn.cpp:38-51: 16 result bits, 64-bit accumulators on LP64 */
static uint64_t ti_sqrt(uint64_t arg)
{
    int i;
    uint64_t nprd,msk=0x8000,val=0,prd=0;
    for(i=15;i>=0;i--)
    {
        nprd=prd+(val<<(i+1))+(msk<<i);
        if(nprd<=arg){ val|=msk; prd=nprd; }
        msk>>=1;
    }
    return val;
}

/* T_norm, TransLn.cpp:75-103 (fixed-point branch); long is 64 bits in the pinned build */
static i64 t_norm(const i32 *v)
{
    if((v[0]>0xffff)||(v[1]>0xffff)||(v[2]>0xffff)||(v[0]<-0xffff)||(v[1]<-0xffff)||(v[2]<-0xffff))
        return (i64)(ti_sqrt((uint64_t)((((i64)v[0])>>8)*v[0]+(((i64)v[1])>>8)*v[1]+(((i64)v[2])>>8)*v[2]))<<4);
    return (i64)ti_sqrt((uint64_t)(((i64)v[0])*v[0]+((i64)v[1])*v[1]+((i64)v[2])*v[2]));
}

/* T_unit_vector, TransLn.cpp:175-190; returns 0 where the reference stops with
 * "Found a zero length vector" */
static int unit_vector(const i32 *from1,const i32 *from2,i32 *to)
{
    i32 v[3];
    i64 lng;
    v[0]=wsub(from2[0],from1[0]); v[1]=wsub(from2[1],from1[1]); v[2]=wsub(from2[2],from1[2]);
    lng=t_norm(v);
    if(lng==0) return 0;
    /* (long)v << 7 as the reference writes it: a multiplication keeps negative values defined */
    to[0]=(i32)((((i64)v[0])*(1<<LTO_LOG_NORMAL))/lng);
    to[1]=(i32)((((i64)v[1])*(1<<LTO_LOG_NORMAL))/lng);
    to[2]=(i32)((((i64)v[2])*(1<<LTO_LOG_NORMAL))/lng);
    return 1;
}

/* T_scalar_product, TransLn.cpp:201-208 */
static i32 scalar_product(const i32 *a,const i32 *b)
{
    return (i32)((((i64)a[0]*(i64)b[0])+((i64)a[1]*(i64)b[1])+((i64)a[2]*(i64)b[2]))>>LTO_LOG_NORMAL);
}

/* M_shade_vertex, EngBase.cpp:220-288 (_RGB_); returns the number of lights skipped
 * because the reference would have stopped on a zero-length vector */
static int shade_vertex(i32 *rgb,const i32 *vertex,const i32 *normal,int nl,const i32 *lights)
{
    int j,skipped=0;
    rgb[0]=rgb[1]=rgb[2]=0;
    for(j=0;j<nl;j++)
    {
        const i32 *l=lights+j*8;
        i32 lv[3],prd;
        if(l[0]==1){ rgb[0]=wadd(rgb[0],l[5]); rgb[1]=wadd(rgb[1],l[6]); rgb[2]=wadd(rgb[2],l[7]); }
        else if(l[0]==2)
        {
            if(!unit_vector(vertex,l+1,lv)){ skipped++; continue; }
            prd=scalar_product(lv,normal);
            if(prd<0) continue;
            rgb[0]=wadd(rgb[0],wmul(prd,l[5])>>LTO_LOG_NORMAL);
            rgb[1]=wadd(rgb[1],wmul(prd,l[6])>>LTO_LOG_NORMAL);
            rgb[2]=wadd(rgb[2],wmul(prd,l[7])>>LTO_LOG_NORMAL);
        }
        else if(l[0]==4)
        {
            prd=(i32)(0u-(u32)scalar_product(l+1,normal));
            if(prd<0) continue;
            rgb[0]=wadd(rgb[0],wmul(prd,l[5])>>LTO_LOG_NORMAL);
            rgb[1]=wadd(rgb[1],wmul(prd,l[6])>>LTO_LOG_NORMAL);
            rgb[2]=wadd(rgb[2],wmul(prd,l[7])>>LTO_LOG_NORMAL);
        }
    }
    return skipped;
}

/* M_shade_polygon_object + M_shade_polygon (EngPoly.cpp:464-493, EngBase.cpp:325-364).
 * s9: self rotation; (x,y,z): object position.  verts[nv][4], normals[nn][4];
 * polys[np][4] = {colour, tex, n_edges, first}; polyverts[*][6]: R,G,B of every record of
 * every polygon are overwritten; polynormals[*]: normal index per record (m_normals);
 * polyaux[np][4]: m_red, m_green, m_blue overwritten with the flat illumination.
 * Returns the number of (vertex, light) pairs skipped for a zero-length light vector;
 * polygons whose records, vertex or normal indices are out of range are left alone and
 * counted in *bad_polys. */
int lto_shade_polys(const i32 *s9,i32 x,i32 y,i32 z,const i32 *verts,int nv,const i32 *normals,int nn,
                    const i32 *polys,int np,i32 *polyverts,const i32 *polynormals,int npv,i32 *polyaux,
                    const i32 *lights,int nl,int *bad_polys)
{
    int i,k,skipped=0,bad=0;
    for(i=0;i<np;i++)
    {
        int e=polys[i*4+2],ok=1;
        i64 first=polys[i*4+3];
        i32 sum[3]={0,0,0};
        if(e<0||first<0||first+e+1>npv) ok=0;
        for(k=0;ok&&k<=e;k++)
        {
            i32 vi=polyverts[(first+k)*6],ni=polynormals[first+k];
            if(vi<0||vi>=nv||ni<0||ni>=nn) ok=0;
        }
        if(!ok){ bad++; continue; }
        for(k=0;k<=e;k++)
        {
            i32 *rec=polyverts+(first+k)*6,wv[3],wn[3];
            self_rotate(s9,verts+(size_t)rec[0]*4,wv);
            wv[0]=wadd(wv[0],x); wv[1]=wadd(wv[1],y); wv[2]=wadd(wv[2],z);      /* T_translation, TransBs.cpp:86-100 */
            self_rotate(s9,normals+(size_t)polynormals[first+k]*4,wn);
            skipped+=shade_vertex(rec+1,wv,wn,nl,lights);
            sum[0]=wadd(sum[0],rec[1]); sum[1]=wadd(sum[1],rec[2]); sum[2]=wadd(sum[2],rec[3]);
        }
        polyaux[i*4+0]=tdiv(sum[0],e+1);
        polyaux[i*4+1]=tdiv(sum[1],e+1);
        polyaux[i*4+2]=tdiv(sum[2],e+1);
    }
    if(bad_polys) *bad_polys=bad;
    return skipped;
}

/* ------------------------------------------------------------------ normals (object set-up) */

/* T_normal_plane, TransLn.cpp:150-171: unit normal (T_LOG_NORMAL_SIZE fraction bits) of the
 * plane through three points; products wrap in 32 bits as the reference's int arithmetic
 * does.  Returns 0 where the reference stops with "Found a zero length vector". */
int lto_normal_plane(const i32 *from1,const i32 *from2,const i32 *from3,i32 *to)
{
    i32 v[3],w[3],n[3];
    i64 lng;
    v[0]=wsub(from2[0],from1[0]); v[1]=wsub(from2[1],from1[1]); v[2]=wsub(from2[2],from1[2]);
    w[0]=wsub(from3[0],from1[0]); w[1]=wsub(from3[1],from1[1]); w[2]=wsub(from3[2],from1[2]);
    n[0]=wsub(wmul(v[1],w[2]),wmul(v[2],w[1]));
    n[1]=wsub(wmul(v[2],w[0]),wmul(v[0],w[2]));
    n[2]=wsub(wmul(v[0],w[1]),wmul(v[1],w[0]));
    lng=t_norm(n);
    if(lng==0) return 0;
    to[0]=(i32)((((i64)n[0])*(1<<LTO_LOG_NORMAL))/lng);
    to[1]=(i32)((((i64)n[1])*(1<<LTO_LOG_NORMAL))/lng);
    to[2]=(i32)((((i64)n[2])*(1<<LTO_LOG_NORMAL))/lng);
    return 1;
}

/* MI_add_tuple, EngPoly.cpp:46-63: index of an X Y Z tuple in the list, appended if new;
 * -1 where the reference stops with "Not enough internal storage for vertices". */
static int add_tuple(i32 *list,int *count,int limit,const i32 *t)
{
    int i;
    for(i=0;i<*count;i++)
        if(list[i*4]==t[0]&&list[i*4+1]==t[1]&&list[i*4+2]==t[2]) return i;
    if(i>=limit) return -1;
    list[i*4]=t[0]; list[i*4+1]=t[1]; list[i*4+2]=t[2];
    return (*count)++;
}

/* The normals half of M_init_polygon_object, EngPoly.cpp:390-458 (no BSP ordering: the
 * pinned build has a depth buffer).  verts[nv][4]; polys[np][4] = {colour, tex, n_edges,
 * first}; types[np] = m_type (M_PLANNAR 1, M_CURVED 2); polyverts[*][6] vertex records.
 * normals[limit][4] receives the list (X Y Z; the fourth word is left as it is),
 * polynormals[*] the index per record, laid out like polyverts (entries of polygons that
 * are neither planar nor curved are left as they are: the reference leaves m_normals NULL).
 * Returns the number of normals, -1 for a zero-length plane normal, -2 for a full list. */
int lto_init_normals(const i32 *verts,int nv,const i32 *polys,const i32 *types,int np,
                     const i32 *polyverts,i32 *normals,int limit,i32 *polynormals)
{
    i32 *plane=(i32*)malloc((size_t)(np>0?np:1)*4*sizeof(i32));
    char *have=(char*)calloc((size_t)(np>0?np:1),1);
    int i,j,k,count=0,rc=0;
    for(i=0;i<np&&rc==0;i++)
    {
        const i32 *pv=polyverts+(size_t)polys[i*4+3]*6;
        i32 *pn=polynormals+polys[i*4+3];
        int e=polys[i*4+2];
        if(!lto_normal_plane(verts+(size_t)pv[0]*4,verts+(size_t)pv[6]*4,verts+(size_t)pv[12]*4,plane+i*4)){ rc=-1; break; }
        if(types[i]&1)
        {
            int idx=add_tuple(normals,&count,limit,plane+i*4);
            if(idx<0){ rc=-2; break; }
            for(j=0;j<=e;j++) pn[j]=idx;
            have[i]=1;
        }
    }
    for(i=0;i<nv&&rc==0;i++)
    {
        i32 n[3]={0,0,0};
        int num=0;
        for(j=0;j<np;j++)
        {
            if(types[j]&2)
            {
                const i32 *pv=polyverts+(size_t)polys[j*4+3]*6;
                i32 *pn=polynormals+polys[j*4+3];
                int e=polys[j*4+2];
                if(!have[j]){ for(k=0;k<=e;k++) pn[k]=0; have[j]=1; }
                for(k=0;k<=e;k++)
                    if(pv[k*6]==i)
                    {
                        pn[k]=-1;
                        n[0]=wadd(n[0],plane[j*4]); n[1]=wadd(n[1],plane[j*4+1]); n[2]=wadd(n[2],plane[j*4+2]);
                        num++;
                    }
            }
        }
        if(num!=0)
        {
            int idx;
            n[0]/=num; n[1]/=num; n[2]/=num;
            idx=add_tuple(normals,&count,limit,n);
            if(idx<0){ rc=-2; break; }
            for(j=0;j<np;j++)
                if(types[j]&2)
                {
                    i32 *pn=polynormals+polys[j*4+3];
                    int e=polys[j*4+2];
                    for(k=0;k<=e;k++) if(pn[k]==-1) pn[k]=idx;
                }
        }
    }
    free(plane); free(have);
    return rc<0?rc:count;
}

/* ------------------------------------------------------------------ bicubic patches (EngBcub.cpp)
 *
 * A patch is 16 control points int32[16][4]; it is tessellated into (size-1)^2 squares,
 * size = (1 << log_render_size) + 1, by recursive halving of the Bezier control polygons in
 * integer arithmetic, and every square goes through M_render_polygon with per-point
 * colours and texture coordinates. */

#define LTO_M_PB 8                           /* Engine.h:78 */
#define LTO_MAX_RENDER_SIZE 33               /* Engine.h:275 */

static i32 half_sum(i32 a,i32 b){ return tdiv(wadd(a,b),2); }

/* MI_evaluate_bezier_points, EngBcub.cpp:113-152; the top-level call with `rendering` shifts
 * its four controls left by M_PB in place, as the reference does */
static void bezier_points(i32 *b1,i32 *b2,i32 *b3,i32 *b4,i32 *points,int dim,int length,int top,int rendering)
{
    i32 b12[3],b23[3],b34[3],b123[3],b234[3],*b1234;
    int k;
    if(top)
    {
        if(rendering)
            for(k=0;k<3;k++)
            { b1[k]=wshl(b1[k],LTO_M_PB); b2[k]=wshl(b2[k],LTO_M_PB); b3[k]=wshl(b3[k],LTO_M_PB); b4[k]=wshl(b4[k],LTO_M_PB); }
        memcpy(points,b1,3*sizeof(i32));
        memcpy(points+(length-1)*dim,b4,3*sizeof(i32));
    }
    if(length>2)
    {
        b1234=points+(length/2)*dim;
        for(k=0;k<3;k++)
        {
            b12[k]=half_sum(b1[k],b2[k]); b23[k]=half_sum(b2[k],b3[k]); b34[k]=half_sum(b3[k],b4[k]);
            b123[k]=half_sum(b12[k],b23[k]); b234[k]=half_sum(b23[k],b34[k]);
            b1234[k]=half_sum(b123[k],b234[k]);
        }
        bezier_points(b1,b12,b123,b1234,points,dim,length/2+1,0,rendering);
        bezier_points(b1234,b234,b34,b4,b1234,dim,length/2+1,0,rendering);
    }
}

/* MI_evaluate_bezier_tangents, EngBcub.cpp:163-196 */
static void bezier_tangents(const i32 *b1,const i32 *b2,const i32 *b3,const i32 *b4,i32 *tangents,int dim,int length,int top)
{
    i32 b12[3],b23[3],b34[3],b123[3],b234[3],b1234[3],*mid;
    int k;
    if(top)
        for(k=0;k<3;k++)
        { tangents[k]=wsub(b2[k],b1[k]); tangents[(length-1)*dim+k]=wsub(b4[k],b3[k]); }
    if(length>2)
    {
        mid=tangents+(length/2)*dim;
        for(k=0;k<3;k++)
        {
            b12[k]=half_sum(b1[k],b2[k]); b23[k]=half_sum(b2[k],b3[k]); b34[k]=half_sum(b3[k],b4[k]);
            b123[k]=half_sum(b12[k],b23[k]); b234[k]=half_sum(b23[k],b34[k]);
            mid[k]=wsub(b234[k],b123[k]);
            b1234[k]=half_sum(b123[k],b234[k]);
        }
        bezier_tangents(b1,b12,b123,b1234,tangents,dim,length/2+1,0);
        bezier_tangents(b1234,b234,b34,b4,mid,dim,length/2+1,0);
    }
}

/* the four curves across the patch, then one curve per column: points[size][size][4] */
static void patch_points(i32 *c,i32 *points,int size,int rendering)
{
    i32 t1[LTO_MAX_RENDER_SIZE*4],t2[LTO_MAX_RENDER_SIZE*4],t3[LTO_MAX_RENDER_SIZE*4],t4[LTO_MAX_RENDER_SIZE*4];
    int i;
    bezier_points(c,c+16,c+32,c+48,t1,4,size,1,rendering);
    bezier_points(c+4,c+20,c+36,c+52,t2,4,size,1,rendering);
    bezier_points(c+8,c+24,c+40,c+56,t3,4,size,1,rendering);
    bezier_points(c+12,c+28,c+44,c+60,t4,4,size,1,rendering);
    for(i=0;i<size;i++)
        bezier_points(t1+i*4,t2+i*4,t3+i*4,t4+i*4,points+(size_t)i*size*4,4,size,1,0);
}

/* M_init_bicubic, EngBcub.cpp:204-265: normals[size*size][4] (three words each are written).
 * Returns 0, -1 where the reference stops on a zero-length normal, -2 for a size it refuses. */
int lto_init_bicubic(const i32 *controls,int log_render_size,i32 *normals)
{
    static i32 tan1[LTO_MAX_RENDER_SIZE*LTO_MAX_RENDER_SIZE*4],tan2[LTO_MAX_RENDER_SIZE*LTO_MAX_RENDER_SIZE*4];
    i32 c[64],t1[LTO_MAX_RENDER_SIZE*4],t2[LTO_MAX_RENDER_SIZE*4],t3[LTO_MAX_RENDER_SIZE*4],t4[LTO_MAX_RENDER_SIZE*4];
    static const i32 origin[3]={0,0,0};
    int i,size=(1<<log_render_size)+1;
    if(log_render_size<0||size>=LTO_MAX_RENDER_SIZE) return -2;
    memcpy(c,controls,sizeof c);
    bezier_points(c,c+4,c+8,c+12,t1,4,size,1,0);
    bezier_points(c+16,c+20,c+24,c+28,t2,4,size,1,0);
    bezier_points(c+32,c+36,c+40,c+44,t3,4,size,1,0);
    bezier_points(c+48,c+52,c+56,c+60,t4,4,size,1,0);
    for(i=0;i<size;i++) bezier_tangents(t1+i*4,t2+i*4,t3+i*4,t4+i*4,tan1+i*4,4*size,size,1);
    bezier_points(c,c+16,c+32,c+48,t1,4,size,1,0);
    bezier_points(c+4,c+20,c+36,c+52,t2,4,size,1,0);
    bezier_points(c+8,c+24,c+40,c+56,t3,4,size,1,0);
    bezier_points(c+12,c+28,c+44,c+60,t4,4,size,1,0);
    for(i=0;i<size;i++) bezier_tangents(t1+i*4,t2+i*4,t3+i*4,t4+i*4,tan2+(size_t)i*size*4,4,size,1);
    for(i=0;i<size*size;i++)                 /* T_normal_vectors = T_normal_plane from the origin */
        if(!lto_normal_plane(origin,tan1+i*4,tan2+i*4,normals+i*4)) return -1;
    return 0;
}

/* M_shade_bicubic, EngBcub.cpp:271-326: intensities[size*size][3]; returns the number of
 * (point, light) pairs skipped for a zero-length light vector, -2 for a refused size. */
int lto_shade_bicubic(const i32 *s9,i32 x,i32 y,i32 z,const i32 *controls,int log_render_size,const i32 *normals,
                      const i32 *lights,int nl,i32 *intensities)
{
    static i32 points[LTO_MAX_RENDER_SIZE*LTO_MAX_RENDER_SIZE*4];
    i32 c[64],wn[3];
    int i,skipped=0,size=(1<<log_render_size)+1;
    if(log_render_size<0||size>LTO_MAX_RENDER_SIZE) return -2;
    for(i=0;i<16;i++)
    {
        self_rotate(s9,controls+i*4,c+i*4);
        c[i*4]=wadd(c[i*4],x); c[i*4+1]=wadd(c[i*4+1],y); c[i*4+2]=wadd(c[i*4+2],z); c[i*4+3]=0;
    }
    patch_points(c,points,size,0);
    for(i=0;i<size*size;i++)
    {
        self_rotate(s9,normals+i*4,wn);
        skipped+=shade_vertex(intensities+i*3,points+i*4,wn,nl,lights);
    }
    return skipped;
}

/* M_render_bicubic, EngBcub.cpp:332-403, after the caller has built the matrix m
 * (T_set_self_rotation + T_cancatinate_self_world): one M_render_polygon per square
 * (MI_render_square, :68-102).  tex = texture handle or -1.  Returns 0, -2 for a refused size,
 * -3 where the reference stops with "Too small a texture to fit onto a bicubic". */
int lto_render_bicubic(lto_ctx *ctx,const i32 *m,const i32 *controls,int log_render_size,i32 colour,
                       int log_texture_space_size,int tex,const i32 *intensities,int render_type,int log_focus,
                       int force_linear)
{
    static i32 points[LTO_MAX_RENDER_SIZE*LTO_MAX_RENDER_SIZE*4];
    static const int corner[5]={2,3,1,0,2};  /* M_square_v, EngBcub.cpp:49-56 */
    i32 c[64];
    int i,k,size=(1<<log_render_size)+1,incr=log_texture_space_size-log_render_size,step,max;
    if(log_render_size<0||size>LTO_MAX_RENDER_SIZE) return -2;
    if(incr<=0) return -3;
    memset(c,0,sizeof c);
    lto_transform(m,controls,c,16);
    step=1<<incr; max=1<<log_texture_space_size;
    patch_points(c,points,size,1);
    ctx->log_focus=log_focus;
    for(i=0;i<size*size;i++)
    {
        int no[4];
        i32 vert[16],poly[4],aux[4],rec[30];
        if(i%size==size-1||i/size==size-1) continue;
        no[0]=i; no[1]=i+1; no[2]=i+size; no[3]=i+size+1;
        for(k=0;k<4;k++)
        {
            vert[k*4]=points[no[k]*4]>>LTO_M_PB; vert[k*4+1]=points[no[k]*4+1]>>LTO_M_PB; vert[k*4+2]=points[no[k]*4+2]>>LTO_M_PB;
            vert[k*4+3]=0;
        }
        for(k=0;k<5;k++)
        {
            int p=no[corner[k]];
            rec[k*6]=corner[k];
            rec[k*6+1]=intensities[p*3]; rec[k*6+2]=intensities[p*3+1]; rec[k*6+3]=intensities[p*3+2];
            rec[k*6+4]=(p%size)*step; rec[k*6+5]=max-(p/size)*step;      /* the grid of :366-373 */
        }
        poly[0]=colour; poly[1]=tex; poly[2]=4; poly[3]=0;
        aux[0]=intensities[no[0]*3]; aux[1]=intensities[no[0]*3+1]; aux[2]=intensities[no[0]*3+2];
        aux[3]=LTO_M_QUAD_XY|(incr<<8);
        render_polygon(ctx,vert,poly,aux,rec,render_type,force_linear);
    }
    return 0;
}
