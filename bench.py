#!/usr/bin/env python3
"""Benchmark of the polygon rasterization path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--workload c2-small|c2-large|c3-lin]

One step = one frame: clear + one object of `--tris` synthetic triangles (transform,
setup, binning, Z-buffered fill), inputs resident in HBM before the timed region.  With
N > 1 (launched by torch.distributed.run, one rank per GPU) every rank renders its own
frames of the same size (frame-parallel, no collective on the data path; weak scaling)
and `value` is the whole job's triangles per second.

Rank 0 prints one JSON line.  `roofline` is for the slowest kernel of the step:
algorithmic bytes of one frame (160 B per triangle + 8 B per pixel) over that kernel's
mean duration from HIP events recorded on the library's stream.  `cpu_baseline` times
the checker on this host's cores: the reference's own code when its prebuilt checker
library for this screen size travelled with the tree, else the C restatement.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0
BYTES_PER_TRI = 160
BYTES_PER_PIXEL = 8


def make_scene(workload, tris):
    from raytracing_amd import scenes as S
    if workload == "c2-small":
        return S.gen_c2(tris, 1920, 1080, 10, radius=8)
    if workload == "c2-large":
        return S.gen_c2(tris, 1920, 1080, 10, radius=128)
    if workload == "c3-lin":
        return S.gen_c3(tris, 3840, 2160, 11, radius=8, equal_lights=True, lin_percent=100)
    raise SystemExit(f"unknown workload {workload}")


def cpu_baseline(sc, passes, want_frame=True):
    """Time the CPU checker on the same scene; returns (dict, colour, depth)."""
    import oracle.port as P
    import oracle.ref as R
    use_ref = os.path.exists(R.so_path(sc.width, sc.height)) and sc.render_type == 4
    vv = P.transform(sc.matrix, sc.verts)
    lit = (0x000000, 0, 0, 0)
    if use_ref:
        r = R.Ref(sc.width, sc.height)
        t0 = time.perf_counter()
        for _ in range(passes):
            r.clear(*lit)
            r.render_polys(vv, sc.polys, sc.polyaux, sc.polyverts, sc.render_type, sc.log_focus, sc.force_linear)
        dt = time.perf_counter() - t0
        c, z = r.cbuf().copy(), r.zbuf().copy()
        kind = "reference"
    else:
        o = P.Oracle(sc.width, sc.height)
        for h, (tex, lg) in enumerate(sc.textures):
            o.set_texture(h, tex, lg)
        t0 = time.perf_counter()
        for _ in range(passes):
            o.clear(*lit)
            o.render_polys(vv, sc.polys, sc.polyaux, sc.polyverts, sc.render_type, sc.log_focus, sc.force_linear)
        dt = time.perf_counter() - t0
        c, z = o.cbuf().copy(), o.zbuf().copy()
        kind = "port"
    base = {"value": round(sc.n_polys * passes / dt / 1e6, 4), "unit": "Mtris/s", "cores": 1, "kind": kind,
            "sample": f"{passes} pass(es) over the full {sc.n_polys}-triangle frame (transform excluded), 1 thread"}
    return base, c, z


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2-small")
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg and the frame check")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    from raytracing_amd import gpu as G

    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    sc = make_scene(args.workload, args.tris)
    dev = torch.device("cuda", local_rank)
    d_verts = torch.from_numpy(sc.verts).to(dev)
    d_polys = torch.from_numpy(sc.polys).to(dev)
    d_pv = torch.from_numpy(sc.polyverts).to(dev)
    d_aux = torch.from_numpy(sc.polyaux).to(dev) if sc.polyaux is not None else None

    ctx = G.Context(sc.width, sc.height, device=local_rank)
    for h, (tex, lg) in enumerate(sc.textures):
        ctx.texture_upload(h, tex, lg)

    def step():
        ctx.clear(0)
        ctx.submit_object_dev(d_verts, d_polys, d_pv, sc.render_type, sc.log_focus, d_polyaux=d_aux,
                              matrix=sc.matrix, force_linear=sc.force_linear, max_edges=3)

    for _ in range(args.warmup):
        step()
    ctx.flush()
    ctx.reset_counters()
    ctx.set_profiling(True)
    ctx.stage_ms()

    def barrier():
        if world > 1:
            dist.barrier()

    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt_max = float(tmax.item())

    stages, submits = ctx.stage_ms()
    counters = ctx.counters()
    ctx.set_profiling(False)

    if rank == 0:
        total_tris = sc.n_polys * args.steps * world
        frame_bytes = BYTES_PER_TRI * sc.n_polys + BYTES_PER_PIXEL * sc.width * sc.height
        dom = max(stages, key=lambda k: stages[k])
        dom_ms = stages[dom] / max(submits, 1)
        achieved = frame_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        out = {
            "metric": "Z-tested shaded Mfragments/s + Mtris/s at 1080p/4K, 1/2/4/8 MI355X",
            "value": round(total_tris / dt_max / 1e6, 3),
            "unit": "Mtris/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {sc.n_polys} Gouraud Z-buffered triangles per frame at "
                                   f"{sc.width}x{sc.height}, one frame per step per GPU"
                       if args.workload.startswith("c2") else
                       f"{args.workload}: {sc.n_polys} textured triangles per frame at {sc.width}x{sc.height}",
                       "parallelism": f"frame-parallel x{world}", "tile": f"{64}x{32}"},
            "mfragments_per_s": round(counters["fragments"] * world / dt_max / 1e6, 3),
            "fragments_per_frame": counters["fragments"] // max(args.steps, 1),
            "stage_ms_per_frame": {k: round(v / max(submits, 1), 4) for k, v in stages.items()},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "algorithmic_bytes_per_launch": frame_bytes,
                         "whole_step_frac": round(frame_bytes / (dt_max / args.steps) / 1e9 / HBM_PEAK_GBS, 5)},
        }
        if not args.no_cpu:
            base, oc, oz = cpu_baseline(sc, passes=2 if sc.n_polys <= 2_000_000 else 1)
            c, z = ctx.readback()
            out["cpu_baseline"] = base
            out["frame_matches_cpu"] = bool(np.array_equal(c, oc) and np.array_equal(z, oz))
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
