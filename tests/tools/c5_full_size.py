"""Configuration 5 at its own size on one GPU (run by hand on the GPU box; results in its header below and in the
design notes): 100M Gouraud triangles at 16384 x 16384, log_focus 13, vertex offsets of +-16 pixels, drawn

  a. as a whole frame,
  b. in the eight row bands of an 8-GPU sort-first split, every band setting up the whole stream
     (what each of eight GPUs would do with replicated input), one band after the other,
  c. the same with the polygons routed to their bands first (lt_polygon_rows_dev + lists by band),
  d. (--table-objects K, in place of b and c) the whole frame with the objects joined K at a time into tables of
     objects, one GPU batch per table (lt_submit_objects_dev): what the host mirror makes of consecutive calls.

The 100M triangles are ten objects of 10M (gen_c2 with seeds 0xC0FFEE05 + k), submitted in turn: the frame of ten
M_render_polygon_object-like calls.  Checks: the bands' frame equals the whole frame (digests), the bands' fragment
counts add up to the whole frame's (unrouted), the routed frame equals it too.

  python tests/tools/c5_full_size.py [--objects 10] [--tris-per-object 10000000] [--frames 3]

Measured (MI355X, this build): see the line the script prints; recorded in the design notes.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--objects", type=int, default=10)
    ap.add_argument("--tris-per-object", type=int, default=10_000_000)
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--size", type=int, default=16384)
    ap.add_argument("--table-objects", type=int, default=0, help="objects per table for step d (0: steps b and c instead)")
    args = ap.parse_args()
    import torch
    from raytracing_amd import bands as B, gpu as G, scenes as S
    W = H = args.size
    lf = 13
    dev = torch.device("cuda", 0)
    objs = []
    t0 = time.time()
    for k in range(args.objects):
        sc = S.gen_c2(args.tris_per_object, W, H, lf, radius=16, seed=0xC0FFEE05 + k)
        objs.append(tuple(torch.from_numpy(a).to(dev) for a in (sc.verts, sc.polys, sc.polyverts)))
        matrix, rt = sc.matrix, sc.render_type
        del sc
        print(f"object {k} generated and uploaded ({time.time() - t0:.0f} s)", flush=True)
    n_total = args.objects * args.tris_per_object
    ctx = G.Context(W, H)

    def draw(select=None):
        ctx.clear(0)
        for k, (v, po, pv) in enumerate(objs):
            p = po if select is None else select[k]
            if p.shape[0]:
                ctx.submit_object_dev(v, p, pv, rt, lf, matrix=matrix, max_edges=3)

    out = {"workload": f"c5: {n_total} Gouraud triangles as {args.objects} objects at {W}x{H}, log_focus {lf}"}
    # a. whole frame
    draw(); ctx.flush()
    ctx.reset_counters()
    t0 = time.perf_counter()
    for _ in range(args.frames):
        draw()
    ctx.flush()
    dt = (time.perf_counter() - t0) / args.frames
    n = ctx.counters()
    c, z = ctx.readback()
    whole = (S.frame_digest(c), S.frame_digest(z))
    frags_whole = n["fragments"] // args.frames
    out["whole_frame"] = {"ms_per_frame": round(dt * 1e3, 3), "mtris_per_s": round(n_total / dt / 1e6, 1),
                          "mfragments_per_s": round(frags_whole / dt / 1e6, 1), "fragments": frags_whole,
                          "polys_drawn": n["polys_drawn"] // args.frames, "colour": whole[0], "depth": whole[1],
                          "drawn_pixels": int((z != 2**31 - 1).sum())}
    print(json.dumps(out["whole_frame"]), flush=True)
    del c, z
    if args.table_objects > 0:
        # d. tables of objects: the arrays laid end to end, a matrix and a vertex range per object
        K = args.table_objects
        tables = []
        for a0 in range(0, len(objs), K):
            grp = objs[a0:a0 + K]
            tab = np.zeros((len(grp), 16), np.int32)
            polys, vbase, pvbase = [], 0, 0
            for i, (v, po, pv) in enumerate(grp):
                tab[i, :12] = np.asarray(matrix, np.int32).reshape(12)
                tab[i, 12], tab[i, 13] = vbase, v.shape[0]
                q = po.clone()
                q[:, 3] += pvbase
                polys.append(q)
                vbase += v.shape[0]
                pvbase += pv.shape[0]
            owner = torch.cat([torch.full((po.shape[0],), i, dtype=torch.int32, device=dev) for i, (_, po, _) in enumerate(grp)])
            tables.append((torch.from_numpy(tab).to(dev), owner, torch.cat([g[0] for g in grp]), torch.cat(polys), torch.cat([g[2] for g in grp])))
            del polys
        objs.clear()
        torch.cuda.empty_cache()
        ctx.set_list_capacity(int(tables[0][3].shape[0]) * 4)

        def draw_tables():
            ctx.clear(0)
            for (tab, owner, v, po, pv) in tables:
                ctx.submit_objects_dev(tab, owner, v, po, pv, rt, lf, max_edges=3)

        draw_tables(); ctx.flush()
        ctx.reset_counters()
        t0 = time.perf_counter()
        for _ in range(args.frames):
            draw_tables()
        ctx.flush()
        dt = (time.perf_counter() - t0) / args.frames
        n = ctx.counters()
        c, z = ctx.readback()
        got = (S.frame_digest(c), S.frame_digest(z))
        out["tables"] = {"objects_per_table": K, "ms_per_frame": round(dt * 1e3, 3), "mtris_per_s": round(n_total / dt / 1e6, 1),
                         "mfragments_per_s": round(n["fragments"] // args.frames / dt / 1e6, 1),
                         "fragments_equal_whole_frame": n["fragments"] // args.frames == frags_whole,
                         "frame_equals_whole_frame": got == whole}
        print(json.dumps(out["tables"]), flush=True)
        ctx.close()
        print(json.dumps(out), flush=True)
        return
    # b. eight bands, replicated input
    bands = list(B.all_bands(8, H))
    ctx.reset_counters()
    per_band_ms = []
    t_all = time.perf_counter()
    for (y0, y1) in bands:
        ctx.set_band(y0, y1)
        t0 = time.perf_counter()
        draw(); ctx.flush()
        per_band_ms.append((time.perf_counter() - t0) * 1e3)
    t_all = time.perf_counter() - t_all
    n = ctx.counters()
    ctx.set_band(0, H)
    c, z = ctx.readback()
    got = (S.frame_digest(c), S.frame_digest(z))
    out["eight_bands"] = {"ms_per_band": [round(x, 3) for x in per_band_ms], "ms_slowest_band": round(max(per_band_ms), 3),
                          "ms_all_bands_on_one_gpu": round(t_all * 1e3, 3), "fragments": n["fragments"],
                          "fragments_equal_whole_frame": n["fragments"] == frags_whole, "frame_equals_whole_frame": got == whole}
    print(json.dumps(out["eight_bands"]), flush=True)
    del c, z
    # c. routed: rows of every rank's share, lists by band, then each band sets up only its polygons
    t0 = time.perf_counter()
    select = [[] for _ in bands]
    for (v, po, pv) in objs:
        per_band = [[] for _ in bands]
        for k in range(8):
            lo, hi = B.share(k, 8, po.shape[0])
            rows = ctx.polygon_rows_dev(v, po[lo:hi], pv, lf, matrix=matrix, max_edges=3)
            for r, l in enumerate(B.lists_by_band(rows, lo, 8, H)):
                per_band[r].append(l)
        for r in range(len(bands)):
            select[r].append(po[torch.cat(per_band[r]).long()].contiguous())
    torch.cuda.synchronize()
    route_ms = (time.perf_counter() - t0) * 1e3
    ctx.reset_counters()
    per_band_ms = []
    for r, (y0, y1) in enumerate(bands):
        ctx.set_band(y0, y1)
        t0 = time.perf_counter()
        draw(select[r]); ctx.flush()
        per_band_ms.append((time.perf_counter() - t0) * 1e3)
    n = ctx.counters()
    ctx.set_band(0, H)
    c, z = ctx.readback()
    got = (S.frame_digest(c), S.frame_digest(z))
    out["eight_bands_routed"] = {"ms_per_band": [round(x, 3) for x in per_band_ms], "ms_slowest_band": round(max(per_band_ms), 3),
                                 "ms_routing_all_ranks_on_one_gpu": round(route_ms, 3),
                                 "polys_per_band": [int(sum(p.shape[0] for p in s)) for s in select],
                                 "fragments": n["fragments"], "fragments_equal_whole_frame": n["fragments"] == frags_whole,
                                 "frame_equals_whole_frame": got == whole}
    print(json.dumps(out["eight_bands_routed"]), flush=True)
    ctx.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
