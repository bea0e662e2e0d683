"""Runs the step lists of tests/golden/make_golden.py (wire_cases, window_cases) on a GPU context."""
import numpy as np


def run_gpu(P, ctx, steps, device_arrays=False):
    import torch
    for st in steps:
        if st[0] == "clear":
            ctx.clear(P.cl_light(st[1], 255, 255, 255))
        elif st[0] == "window":
            ctx.set_clip_window(*st[1])
        elif st[0] == "gpolys":
            ctx.submit_screen_polys(st[1], st[2])
        elif st[0] == "object":
            _, sc, rt, obj = st
            for h, (tex, lg) in enumerate(sc.textures):
                ctx.texture_upload(h, tex, lg)
            m = P.make_matrix((0, 0, 0, 0, 0, 0), obj, 0)
            if device_arrays:
                d = [None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
                     for a in (sc.verts, sc.polys, sc.polyverts, sc.polyaux)]
                ctx.submit_object_dev(d[0], d[1], d[2], rt, sc.log_focus, d_polyaux=d[3], matrix=m)
                ctx.flush()
            else:
                ctx.submit_object(sc.verts, sc.polys, sc.polyverts, rt, sc.log_focus, polyaux=sc.polyaux, matrix=m)
        else:
            l = st[1].copy()
            l[:, 6] = [int(c) if k == 2 else P.cl_light(int(c), 255, 255, 255) for c, k in zip(l[:, 6], l[:, 7])]
            if device_arrays:
                d = torch.from_numpy(l).cuda()
                ctx.submit_lines_dev(d)
                ctx.flush()
            else:
                ctx.submit_lines(l)
    return ctx.readback()
