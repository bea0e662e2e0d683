/* Headless stand-in for the reference's Win32 platform header, used ONLY when the
 * reference sources are compiled into the checker library under oracle/_ref/.
 *
 * It provides what LightTrack.h:13-52,68 of the reference provides (memory
 * helpers, the HW_* integer typedefs, the pixel type for _32BPP_, the screen
 * geometry macros and the HW_error prototype) without <windows.h>, and takes
 * the screen size from the compiler command line instead of the hard-wired
 * 600x600 (LightTrack.h:44-45).  Test infrastructure, never linked into the
 * product library.
 */
#ifndef LT_SHADOW_LIGHTTRACK_H
#define LT_SHADOW_LIGHTTRACK_H

#include <string.h>

#define HW_set_char(dst,lng,val)  memset(dst,val,lng)
#define HW_copy_char(src,dst,lng) memcpy(dst,src,lng)

void HW_set_int(int *dst,long lng,int val);
#define HW_copy_int(src,dst,lng)  memcpy(dst,src,(lng)*sizeof(int))

typedef char HW_8_bit;
#define HW_set_8_bit(dst,lng,val) HW_set_char(dst,lng,val)
#define HW_copy_8_bit(src,dst,lng) HW_copy_char(src,dst,lng)

typedef short HW_16_bit;
void HW_set_16_bit(HW_16_bit *dst,long lng,HW_16_bit val);
#define HW_copy_16_bit(src,dst,lng) memcpy(dst,src,(lng)*sizeof(HW_16_bit))

typedef int HW_32_bit;
#define HW_set_32_bit(dst,lng,val) HW_set_int(dst,lng,val)
#define HW_copy_32_bit(src,dst,lng) HW_copy_int(src,dst,lng)

#if defined(_32BPP_)
typedef HW_32_bit HW_pixel;
#define HW_set_pixel(dst,lng,val) HW_set_32_bit(dst,lng,val)
#define HW_copy_pixel(src,dst,lng) HW_copy_32_bit(src,dst,lng)
#else
#error the checker build is _32BPP_ only
#endif

#ifndef HW_SCREEN_X_SIZE
#error pass -DHW_SCREEN_X_SIZE=<W> -DHW_SCREEN_Y_SIZE=<H>
#endif
#if defined(_MONO_)
 #define HW_SCREEN_LINE_SIZE HW_SCREEN_X_SIZE
#endif
#define HW_SCREEN_X_MAX    (HW_SCREEN_X_SIZE-1)
#define HW_SCREEN_Y_MAX    (HW_SCREEN_Y_SIZE-1)
#define HW_SCREEN_X_CENTRE (HW_SCREEN_X_SIZE/2)
#define HW_SCREEN_Y_CENTRE (HW_SCREEN_Y_SIZE/2)

void HW_error(char *s,...);

#endif
