"""Run by hand on the GPU box: the benchmark's one million triangles through the C++ mirror of the reference's API
(raytracing_amd/host), as one M_render_polygon_object call and as 4000 calls of 250 triangles (the reference's own
limit is 256 polygons a call, Engine.h:432-433) -- host arrays, so every frame pays the copy to the GPU.

  python tests/tools/mirror_many_objects.py [--tris 1000000] [--per-object 250] [--frames 5]
"""
import argparse
import ctypes as C
import json
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--per-object", type=int, default=250)
    ap.add_argument("--frames", type=int, default=5)
    args = ap.parse_args()
    import torch  # noqa: F401  (HIP runtime first)
    from raytracing_amd import _lib, scenes as S
    _lib.lib()
    L = C.CDLL(os.path.join(os.path.dirname(_lib.SO_PATH), "liblt_engine.so"))
    L.lte_cbuf.restype = C.POINTER(C.c_int32)
    L.lte_time_objects.restype = C.c_double
    sc = S.gen_c2(args.tris, 1920, 1080, 10, radius=8)
    L.lte_init(sc.width, sc.height)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    verts, polys, pv = (np.ascontiguousarray(a, np.int32) for a in (sc.verts, sc.polys, sc.polyverts))
    first = polys[:, 3]
    assert np.array_equal(pv[first, 0], 3 * np.arange(args.tris)), "a soup: three vertices a triangle, in polygon order"
    out = {"workload": f"{args.tris} Gouraud triangles at 1920x1080, host arrays"}
    per = args.per_object
    rel = pv.copy()
    obj_of_record = np.repeat(np.arange(args.tris) // per, 4)          # four records a triangle
    rel[:, 0] -= (obj_of_record * 3 * per).astype(np.int32)
    calls = C.c_double(0)
    ms = L.lte_time_objects(p(verts), verts.shape[0], p(polys), None, p(rel), args.tris, per, sc.render_type, sc.log_focus, args.frames, C.byref(calls))
    c_mirror = np.ctypeslib.as_array(L.lte_cbuf(), shape=(sc.height, sc.width)).copy()
    out["mirror_objects_of_%d" % per] = {"ms_per_frame": round(ms, 3), "ms_inside_the_calls": round(calls.value, 3), "calls_per_frame": (args.tris + per - 1) // per}
    # the same triangles as one object through the C-ABI (M_render_polygon_object itself refuses more than 256 polygons
    # a call, EngPoly.cpp:507-508): host arrays too
    import time
    from raytracing_amd import gpu as G
    ctx = G.Context(sc.width, sc.height)

    def frame():
        ctx.clear(G.cl_light(0, 255, 255, 255) if hasattr(G, "cl_light") else 0)
        ctx.submit_object(sc.verts, sc.polys, sc.polyverts, sc.render_type, sc.log_focus, polyaux=sc.polyaux, matrix=sc.matrix, max_edges=3)
    frame(); ctx.flush()
    t0 = time.perf_counter()
    for _ in range(args.frames):
        frame()
    ctx.flush()
    out["c_abi_one_object"] = {"ms_per_frame": round((time.perf_counter() - t0) / args.frames * 1e3, 3)}
    c_one, _ = ctx.readback()
    out["frames_equal"] = bool(np.array_equal(c_one, c_mirror))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
