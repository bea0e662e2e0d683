"""Loader of libltgpu.so, the HIP library behind include/lt_gpu.h.

There is no CPU fallback: if the library has not been built (see
__graft_entry__.build) importing anything that draws raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libltgpu.so")

EXPORTS = [
    "lt_ctx_create", "lt_ctx_destroy", "lt_last_error", "lt_set_stream", "lt_set_list_capacity",
    "lt_clear", "lt_flush", "lt_readback", "lt_framebuffer", "lt_set_frame_overlap", "lt_get_counters", "lt_reset_counters",
    "lt_texture_upload", "lt_transform", "lt_submit_object", "lt_submit_object_dev",
    "lt_submit_screen_polys", "lt_set_profiling", "lt_get_stage_ms", "lt_set_textured_order",
    "lt_bind_framebuffer", "lt_set_band", "lt_set_pipelining", "lt_shade_object_dev", "lt_shade_object",
    "lt_plane_normals_dev", "lt_plane_normals", "lt_submit_lines_dev", "lt_submit_lines",
    "lt_set_clip_window", "lt_shade_points_dev", "lt_shade_points",
    "lt_set_tile_width", "lt_submit_object_dev_noalloc", "lt_submit_screen_polys_prp", "lt_polygon_rows_dev",
    "lt_submit_objects", "lt_submit_objects_dev", "lt_set_tile_waves",
    "lt_set_copy_threads", "lt_host_alloc", "lt_host_free",
]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "fragments", "polys_in", "polys_drawn", "tile_refs", "overflow_polys",
        "order_dependent_rows", "unsupported_polys", "list_overflow", "zero_light_vectors")]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            f"{SO_PATH} is missing: build the HIP library first "
            "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
    # torch ships its own HIP/HSA runtime under the same sonames as /opt/rocm.  Two
    # runtimes in one process cannot both own the GPU, so torch is loaded first and the
    # library then binds to the runtime that is already in the process.
    import torch  # noqa: F401
    L = C.CDLL(SO_PATH)
    vp, i32, u64 = C.c_void_p, C.c_int, C.c_uint64
    L.lt_ctx_create.argtypes = [i32, i32, i32, C.POINTER(vp)]
    L.lt_ctx_destroy.argtypes = [vp]
    L.lt_last_error.argtypes = [vp]
    L.lt_last_error.restype = C.c_char_p
    L.lt_set_stream.argtypes = [vp, vp]
    L.lt_set_copy_threads.argtypes = [vp, i32]
    L.lt_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.lt_host_free.argtypes = [vp]
    L.lt_set_list_capacity.argtypes = [vp, u64]
    L.lt_clear.argtypes = [vp, C.c_int32]
    L.lt_flush.argtypes = [vp]
    L.lt_readback.argtypes = [vp, vp, vp]
    L.lt_framebuffer.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.lt_bind_framebuffer.argtypes = [vp, vp, vp]
    L.lt_set_band.argtypes = [vp, i32, i32]
    L.lt_shade_object_dev.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp, i32, vp, i32, vp, vp, i32, vp, vp, i32]
    L.lt_shade_object.argtypes = L.lt_shade_object_dev.argtypes
    L.lt_plane_normals_dev.argtypes = [vp, vp, i32, vp, i32, vp, i32, vp]
    L.lt_plane_normals.argtypes = L.lt_plane_normals_dev.argtypes
    L.lt_submit_lines_dev.argtypes = [vp, vp, i32]
    L.lt_submit_lines.argtypes = [vp, vp, i32]
    L.lt_set_clip_window.argtypes = [vp, i32, i32, i32, i32]
    L.lt_set_tile_width.argtypes = [vp, i32]
    L.lt_set_tile_waves.argtypes = [vp, i32]
    L.lt_shade_points_dev.argtypes = [vp, vp, vp, vp, i32, vp, i32, vp]
    L.lt_shade_points.argtypes = L.lt_shade_points_dev.argtypes
    L.lt_set_pipelining.argtypes = [vp, i32]
    L.lt_set_frame_overlap.argtypes = [vp, i32]
    L.lt_get_counters.argtypes = [vp, C.POINTER(Counters)]
    L.lt_reset_counters.argtypes = [vp]
    L.lt_texture_upload.argtypes = [vp, i32, vp, i32]
    L.lt_transform.argtypes = [vp, vp, vp, vp, i32]
    L.lt_submit_object.argtypes = [vp, vp, vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, i32]
    L.lt_submit_object_dev.argtypes = [vp, vp, vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, i32]
    L.lt_submit_object_dev_noalloc.argtypes = L.lt_submit_object_dev.argtypes
    L.lt_submit_objects.argtypes = [vp, vp, i32, vp, vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, i32]
    L.lt_submit_objects_dev.argtypes = L.lt_submit_objects.argtypes
    L.lt_polygon_rows_dev.argtypes = [vp, vp, vp, i32, vp, i32, vp, i32, i32, i32, vp]
    L.lt_submit_screen_polys.argtypes = [vp, i32, vp, i32, vp, i32, i32]
    L.lt_submit_screen_polys_prp.argtypes = [vp, vp, i32, vp, vp, i32, i32, i32]
    L.lt_set_profiling.argtypes = [vp, i32]
    L.lt_set_textured_order.argtypes = [vp, i32]
    L.lt_get_stage_ms.argtypes = [vp, C.POINTER(C.c_double * 5), C.POINTER(i32)]
    for name in EXPORTS:
        if name != "lt_last_error":
            getattr(L, name).restype = i32
    _lib = L
    return L
