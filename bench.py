#!/usr/bin/env python3
"""Benchmark of the polygon rasterization path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--workload c2-small|c2-large|c3|c3-q2|c3-lin|c4|c5]
                  [--split frames|bands]

With --gpus N > 1 and no launcher around it (WORLD_SIZE unset) the script starts its own N
ranks through torch.distributed.run before anything touches a GPU, and passes their output on.

One step = one frame: clear + one object of `--tris` synthetic triangles (transform,
setup, binning, Z-buffered fill), inputs resident in HBM before the timed region.  With
N > 1 (launched by torch.distributed.run, one rank per GPU) every rank renders its own
frames of the same size (frame-parallel, no collective on the data path; weak scaling)
and `value` is the whole job's triangles per second.  `--split bands` is the sort-first
mode for one huge frame: every rank sets up the same triangles, draws its band of rows,
and rank 0 gathers the bands (point-to-point over RCCL) inside the timed step; the work is
fixed, so that mode reports strong scaling.

Rank 0 prints one JSON line.  `roofline` is for the slowest kernel of the step:
algorithmic bytes of one frame (160 B per triangle + 8 B per pixel) over that kernel's
mean duration from HIP events recorded on the library's stream.  `cpu_baseline` times
the checker on this host's cores: the reference's own code when its prebuilt checker
library for this screen size travelled with the tree, else the C restatement.
`large_class` (default workload) adds the metric's second triangle class, a few C2-large
frames drawn after the headline measurement; `c4` a run of the animated batch (per-frame
matrix); `c3_4k` (one GPU) the 4K textured, clipped workload and its draw-order variant;
`default_hw_queues` the headline step with HIP's default of four hardware queues.

Before the W warm-up steps the script draws `--wake-frames` untimed frames (100 by default, 12 ms;
`config.device_wake_frames`): the CPU legs that run first leave the GPU idle for seconds, and W = 5
frames of 0.1 ms do not bring it out of its idle power state.  The timed region is still exactly K steps
between two barriers + device synchronisations, directly after the W warm-up steps.
"""
import argparse
import json
import os
import socket
import subprocess
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
    if workload == "c4":
        # the animated batch draws the C2-small object under a per-frame matrix (see c4_matrix)
        return S.gen_c2(tris, 1920, 1080, 10, radius=8)
    if workload == "c3":
        # ~1/3 of the triangles meet a screen border or the near plane, 10 % take the perspective
        # mapper, one light value per triangle; the few random texture mappings per million on
        # which the reference itself divides by zero (TransBs.cpp:505) are replaced
        return S.gen_c3(tris, 3840, 2160, 11, radius=8, equal_lights=True, lin_percent=90, regular_mapping=True)
    if workload == "c3-q2":
        # the same with differing vertex lights: light depends on each row's draw history
        return S.gen_c3(tris, 3840, 2160, 11, radius=8, equal_lights=False, lin_percent=90, regular_mapping=True)
    if workload == "c3-lin":
        return S.gen_c3(tris, 3840, 2160, 11, radius=8, equal_lights=True, lin_percent=100, regular_mapping=True)
    if workload == "c5":
        return S.gen_c2(tris, 16384, 16384, 13, radius=16, seed=0xC0FFEE05)
    raise SystemExit(f"unknown workload {workload}")


STAGE_KERNEL = {"fill": "raster_kernel<4, 64", "setup": "setup_fast_kernel", "bin": "scatter_kernel", "scan": "count_kernel"}
PMC_FILE = os.path.join(ROOT, "profiles", "r3_pmc_c2small.json")
KERNEL_SOURCES = ["lt_kernels.hip", "lt_api.hip", "lt_kernels.h", "lt_math.h"]


def kernel_source_sha():
    """sha256 over the sources the HIP library is built from (the counter passes under profiles/ record it:
    counters quoted in the line must belong to the code that is being timed)"""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "raytracing_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def load_pmc(workload, tris):
    """the committed counter passes, or (None, why) when they are absent, of another workload, or of other sources"""
    if workload != "c2-small" or tris != 1_000_000:
        return None, "the committed counter passes are of the c2-small frame of 1M triangles"
    if not os.path.exists(PMC_FILE):
        return None, f"{os.path.relpath(PMC_FILE, ROOT)} is absent"
    pmc = json.load(open(PMC_FILE))
    want = pmc.get("_meta", {}).get("source_sha256")
    have = kernel_source_sha()
    if want != have:
        return None, (f"{os.path.relpath(PMC_FILE, ROOT)} was collected on other kernel sources (sha256 {str(want)[:12]}.. there, "
                      f"{have[:12]}.. here): run profiles/collect.sh again")
    return pmc, None


def measured_traffic(workload, tris, stage):
    """HBM-side bytes per launch of the stage's main kernel from the committed counter passes
    (profiles/: separate `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` runs of this
    benchmark, KiB per launch).  Only for the workload the passes were taken on."""
    pmc, why = load_pmc(workload, tris)
    if pmc is None:
        return None, why
    for name, ctrs in pmc.items():
        if name.startswith("_"):
            continue
        if STAGE_KERNEL.get(stage, "?") in name and "FETCH_SIZE" in ctrs and "WRITE_SIZE" in ctrs:
            f, w = ctrs["FETCH_SIZE"]["mean_per_launch"], ctrs["WRITE_SIZE"]["mean_per_launch"]
            head = f"rocprofv3 FETCH_SIZE {f:.0f} KiB + WRITE_SIZE {w:.0f} KiB per launch of {name.split('(')[0]}; "
            if stage == "setup":
                # 16-byte-per-lane streaming reads: gfx950 reports half their bytes
                return int((2 * f + w) * 1024), head + "FETCH_SIZE doubled (16-byte-per-lane streaming reads are counted at half)"
            return int((f + w) * 1024), head + ("FETCH_SIZE as reported (this kernel gathers 16-byte pieces, a width the gfx950 "
                                                "halving rule is not calibrated for; it counts memory-side requests, Infinity "
                                                "Cache hits included)")
    return None, None


# cycles one SIMD needs per wave64 instruction, by class, with two or more waves resident: measured on
# the GPU by profiles/tools/issue_rate.hip (profiles/r3_issue_rate.jsonl).  The classes are those of the
# SQ_INSTS_VALU_* counters; "int32_simple" are the two-operand add / sub / shift / logic / compare /
# select forms, "int32_other" everything else the INT32 counter takes (multiplies, three-operand and
# SDWA forms), split by the static share of the kernel's disassembly (profiles/tools/valu_mix.py).
VALU_CLASS_CYCLES = {"int32_simple": 2.35, "int32_other": 4.3, "int64": 4.4, "cvt": 4.3, "f64": 4.3, "trans_f32": 8.2,
                     "trans_f64": 16.3, "f32_mul_add": 2.4, "f32_fma": 3.9, "other": 2.6}


def valu_roofline(workload, tris, stage, kernel_ms):
    """Second ceiling for the fill, which is bound by VALU issue: the cycles its wave-instructions need on
    the 1024 SIMDs -- per-class instruction counts of the committed counter passes (SQ_INSTS_VALU_*) times
    the per-class issue cost measured on this GPU model -- over the kernel's measured duration at the
    2.4 GHz peak clock.  frac 1 would be a SIMD that issues a vector instruction in every free slot."""
    pmc, why = load_pmc(workload, tris)
    if pmc is None or kernel_ms <= 0:
        return {"bound": "valu", "kernel": stage, "frac": None, "note": why} if why else None
    for name, ctrs in pmc.items():
        if name.startswith("_") or STAGE_KERNEL.get(stage, "?") not in name or "SQ_INSTS_VALU" not in ctrs:
            continue
        g = lambda k: ctrs.get(k, {}).get("mean_per_launch", 0.0)
        insts = g("SQ_INSTS_VALU")
        mix = pmc.get("_meta", {}).get("valu_mix", {}).get(stage, {})
        simple_share = mix.get("int32_simple_share", 0.5)
        i32 = g("SQ_INSTS_VALU_INT32")
        classes = {"int32_simple": i32 * simple_share, "int32_other": i32 * (1 - simple_share), "int64": g("SQ_INSTS_VALU_INT64"),
                   "cvt": g("SQ_INSTS_VALU_CVT"),
                   "f64": g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_ADD_F64"),
                   "trans_f32": g("SQ_INSTS_VALU_TRANS_F32"), "trans_f64": g("SQ_INSTS_VALU_TRANS_F64"),
                   "f32_mul_add": g("SQ_INSTS_VALU_MUL_F32") + g("SQ_INSTS_VALU_ADD_F32"), "f32_fma": g("SQ_INSTS_VALU_FMA_F32")}
        classes["other"] = max(insts - sum(classes.values()), 0.0)
        have_classes = i32 > 0
        cycles = sum(classes[k] * VALU_CLASS_CYCLES[k] for k in classes) if have_classes else insts * 4.3
        peak_cycles_per_s = 1024 * 2.4e9
        ach = cycles / (kernel_ms * 1e-3)
        return {"bound": "valu", "kernel": stage, "achieved": round(ach / 1e9, 1), "peak": round(peak_cycles_per_s / 1e9, 1),
                "unit": "G SIMD issue cycles/s", "frac": round(ach / peak_cycles_per_s, 4),
                "wave_instructions_per_launch": int(insts),
                "cycles_per_wave_instruction": round(cycles / insts, 3) if insts else None,
                "classes": {k: int(v) for k, v in classes.items()} if have_classes else None,
                "note": "issue cycles = per-class SQ_INSTS_VALU_* counts x cycles per wave64 instruction measured by "
                        "profiles/tools/issue_rate.hip with >= 2 waves a SIMD (most VALU forms take 4.3 cycles here, the simple "
                        "two-operand integer forms 2.35, v_rcp_f32 8.2, v_rcp_f64 16.3)"}
    return None


def cpu_baseline(sc, passes):
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


def cpu_touched_rows(sc):
    """The CPU restatement with the row tables reset only over the rows the last polygon
    touched (the reference resets all H rows per polygon, GraphicsPoly.cpp:209-216): what a
    tuned single-thread CPU build of the same arithmetic does, reported beside the faithful
    figure."""
    import oracle.port as P
    o = P.Oracle(sc.width, sc.height)
    o.set_fast_reset(True)
    for h, (tex, lg) in enumerate(sc.textures):
        o.set_texture(h, tex, lg)
    vv = P.transform(sc.matrix, sc.verts)
    t0 = time.perf_counter()
    o.clear(0x000000, 0, 0, 0)
    o.render_polys(vv, sc.polys, sc.polyaux, sc.polyverts, sc.render_type, sc.log_focus, sc.force_linear)
    dt = time.perf_counter() - t0
    return {"value": round(sc.n_polys / dt / 1e6, 4), "unit": "Mtris/s", "cores": 1, "kind": "port",
            "sample": f"1 pass over the full {sc.n_polys}-triangle frame, 1 thread, row tables reset over touched rows only"}


def large_class(ctx, torch, dev, tris, frames=5):
    """The metric's second triangle class (C2-large: vertex offsets of +-128 pixels) on the same
    context, after the headline measurement: a few frames, host-timed between two flushes."""
    sc = make_scene("c2-large", tris)
    d_verts, d_polys, d_pv = (torch.from_numpy(a).to(dev) for a in (sc.verts, sc.polys, sc.polyverts))

    def frame():
        ctx.clear(0)
        ctx.submit_object_dev(d_verts, d_polys, d_pv, sc.render_type, sc.log_focus, matrix=sc.matrix, max_edges=3)

    # The library picks its tile width from what it drew, so it would draw the first of these
    # frames with the narrow tiles of the headline frames; that one frame is given the wide
    # tiles by hand, so that the narrow fill kernel's launches in a kernel trace of this
    # command are the headline frames only.
    ctx.set_tile_width(128)
    frame()
    ctx.flush()
    ctx.set_tile_width(0)
    frame()
    ctx.flush()
    ctx.reset_counters()
    t0 = time.perf_counter()
    for _ in range(frames):
        frame()
    ctx.flush()
    dt = time.perf_counter() - t0
    frags = ctx.counters()["fragments"]
    return {"workload": f"c2-large: {tris} Gouraud Z-buffered triangles per frame at 1920x1080", "frames": frames,
            "ms_per_frame": round(dt / frames * 1e3, 4), "mtris_per_s": round(tris * frames / dt / 1e6, 3),
            "mfragments_per_s": round(frags / dt / 1e6, 3)}


def cpu_all_cores(workload, tris, cores, passes=3):
    """The same frame on `cores` host cores at once: independent processes (the reference is
    not re-entrant), started before this process touches the GPU.  Returns the dict or None."""
    import subprocess
    start_at = time.time() + 12.0 + tris / 2.5e5           # scene generation and library loading come first
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_worker.py"), workload, str(tris), str(passes), str(start_at)]
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True) for _ in range(cores)]
    rate, kinds = 0.0, set()
    for p in procs:
        out, _ = p.communicate(timeout=600)
        if p.returncode != 0 or not out.strip():
            return None
        secs, kind = out.split()
        rate += tris * passes / float(secs)
        kinds.add(kind)
    return {"value": round(rate / 1e6, 3), "unit": "Mtris/s", "cores": cores, "kind": kinds.pop() if len(kinds) == 1 else "mixed",
            "sample": f"{cores} processes, each {passes} pass(es) over the full {tris}-triangle frame, timed loops started together"}


def bench_shade(args, rank, local_rank, world):
    """--workload shade: the shading pre-pass (M_shade_polygon_object) alone, one call per
    step over `--tris` triangles (four vertex records each), four lights, arrays in HBM."""
    import torch
    import torch.distributed as dist
    from raytracing_amd import gpu as G, scenes as S
    case = S.gen_shading_case(args.tris, 0x5EED, n_verts=3 * args.tris, n_normals=args.tris)
    case["polys"][:, 2] = 3
    case["polys"][:, 3] = np.arange(args.tris, dtype=np.int32) * 4
    case["polyverts"] = case["polyverts"][: 4 * args.tris].copy()
    case["polynormals"] = case["polynormals"][: 4 * args.tris].copy()
    # a triangle soup as in the rasterization workloads: vertices 3p, 3p+1, 3p+2, 3p; one normal per triangle
    tri = np.arange(args.tris, dtype=np.int32)
    case["polyverts"][:, 0] = (3 * tri[:, None] + np.array([0, 1, 2, 0], np.int32)[None, :]).reshape(-1)
    case["polynormals"] = np.repeat(tri, 4).astype(np.int32)
    dev = torch.device("cuda", local_rank)
    d = {k: torch.from_numpy(v).to(dev) for k, v in case.items() if k != "lights"}
    d_aux = torch.zeros((args.tris, 4), dtype=torch.int32, device=dev)
    ctx = G.Context(64, 64, device=local_rank)
    from raytracing_amd.transform import self_matrix
    s9 = self_matrix(31, 200, 99)

    def step():
        ctx.shade_object_dev(s9, (100, -50, 400), d["verts"], d["normals"], d["polys"], d["polyverts"], d["polynormals"],
                             d_aux, case["lights"])

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    if rank == 0:
        recs = 4 * args.tris
        # per record: 24 B record read + 12 B written, 16 B vertex, 16 B normal, 4 B normal index; per polygon 16 + 12 B
        algo = recs * (24 + 12 + 16 + 16 + 4) + args.tris * (16 + 12)
        ms = dt / args.steps * 1e3
        out = {"metric": "shaded vertex records per second (shading pre-pass)", "value": round(recs * world / (ms * 1e-3) / 1e6, 2),
               "unit": "Mrecords/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32/int64", "data": "synthetic",
               "config": {"workload": f"shade: soup of {args.tris} triangles, {recs} vertex records, 4 lights (ambient, 2 point, direct)"},
               "roofline": {"bound": "hbm", "kernel": "shade_kernel", "achieved": round(algo / (ms * 1e-3) / 1e9, 2),
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                            "traffic": None, "algorithmic_bytes_per_launch": algo}}
        if not args.no_cpu:
            import oracle.port as P
            n = min(args.tris, 200_000)
            t0 = time.perf_counter()
            P.shade_polys(s9, (100, -50, 400), case["verts"], case["normals"], case["polys"][:n], case["polyverts"],
                          case["polynormals"], None, case["lights"])
            out["cpu_baseline"] = {"value": round(4 * n / (time.perf_counter() - t0) / 1e6, 3), "unit": "Mrecords/s", "cores": 1,
                                   "kind": "port", "sample": f"the first {n} triangles, 1 thread"}
        print(json.dumps(out), flush=True)


def side_frames(ctx, torch, dist, world, dev, submit, frames, warm=2):
    """`frames` calls of submit(k) between two flushes, bracketed like the headline region
    (barrier + device synchronisation on both sides, max over ranks) -> seconds, fragments"""
    for k in range(warm):
        submit(k)
    try:
        ctx.flush()
    except Exception:            # reported by the caller from the counters
        pass
    ctx.reset_counters()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(frames):
        submit(k)
    err = None
    try:
        ctx.flush()
    except Exception as e:
        err = str(e)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    n = ctx.counters()
    fr = torch.tensor([n["fragments"]], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(fr)
    return float(t.item()), int(fr.item()), n, err


def c4_matrix(f):
    """placement of frame f of the animated batch: T_set_self_rotation(f, 2f, 3f) and
    T_cancatinate_self_world with translation (0, 0, 2048) under an identity camera
    (reference EngPoly.cpp:510-512)"""
    from raytracing_amd import frames as F
    return F.frame_matrix(f)


def c4_side(ctx, torch, dist, rank, world, dev, d_verts, d_polys, d_pv, sc, frames_per_rank=48):
    """configuration 4 beside the headline: this rank's share of the 1024-frame animated
    batch (frames rank, rank + world, ...), the first `frames_per_rank` of them"""
    from raytracing_amd import frames as F
    mine = list(F.frames_for_rank(1024, rank, world))[:frames_per_rank]
    mats = [c4_matrix(f) for f in mine]

    def submit(k):
        ctx.clear(0)
        ctx.submit_object_dev(d_verts, d_polys, d_pv, sc.render_type, sc.log_focus, matrix=mats[k % len(mats)], max_edges=3)

    dt, frags, _, _ = side_frames(ctx, torch, dist, world, dev, submit, len(mine))
    total = len(mine) * world
    return {"workload": f"c4: frames f = rank, rank + {world}, ... of the 1024-frame animated batch ({sc.n_polys} triangles at "
                        f"{sc.width}x{sc.height}, object angles (f, 2f, 3f) & 255, translation (0, 0, 2048), clear per frame)",
            "frames": total, "ms_per_frame_per_gpu": round(dt / len(mine) * 1e3, 4),
            "mtris_per_s": round(sc.n_polys * total / dt / 1e6, 3), "mfragments_per_s": round(frags / dt / 1e6, 3)}


def many_objects_side(ctx, torch, dist, world, dev, sc, d_verts, d_polys, d_pv, per_object=250, frames=24):
    """The headline frame as a world of small objects, the only way a program written against the reference's
    API can draw it (M_render_polygon_object takes at most 256 polygons, Engine.h:432-433): one table of
    objects (lt_submit_objects_dev), every object with a matrix of its own -- and, for comparison, the same
    objects as one lt_submit_object_dev call each."""
    from raytracing_amd import objects as O
    n = sc.n_polys
    nobj = (n + per_object - 1) // per_object
    table = np.zeros((nobj, O.OBJECT_INTS), np.int32)
    table[:, :12] = sc.matrix.reshape(1, 12)
    table[:, 11] += (np.arange(nobj) % 17 - 8).astype(np.int32)          # every object its own depth offset
    table[:, 12] = np.arange(nobj) * (3 * per_object)
    table[:, 13] = np.minimum(3 * per_object, sc.verts.shape[0] - table[:, 12])
    polyobj = (np.arange(n) // per_object).astype(np.int32)
    pv = sc.polyverts.copy()
    pv[:, 0] -= np.repeat(table[polyobj, 12], 4)                          # vertex indices count from the object's first vertex
    d_table, d_polyobj, d_pv_rel = (torch.from_numpy(a).to(dev) for a in (table, polyobj, pv))

    def submit(k):
        ctx.clear(0)
        ctx.submit_objects_dev(d_table, d_polyobj, d_verts, d_polys, d_pv_rel, sc.render_type, sc.log_focus, max_edges=3)

    dt, frags, _, _ = side_frames(ctx, torch, dist, world, dev, submit, frames)
    out = {"workload": f"many_objects: the {n} triangles as {nobj} objects of {per_object}, one table of objects per frame, a matrix per object",
           "frames": frames, "ms_per_frame": round(dt / frames * 1e3, 4), "mtris_per_s": round(n * frames * world / dt / 1e6, 3),
           "mfragments_per_s": round(frags / dt / 1e6, 3)}
    # one batch per object: object i's slices of the same arrays
    polys_rel = sc.polys.copy()
    d_polys_host = torch.from_numpy(polys_rel).to(dev)
    mats = [table[i, :12].copy() for i in range(nobj)]

    def submit_each(k):
        ctx.clear(0)
        for i in range(nobj):
            a, b = i * per_object, min((i + 1) * per_object, n)
            ctx.submit_object_dev(d_verts[3 * a:3 * b], d_polys_host[a:b], d_pv_rel, sc.render_type, sc.log_focus, matrix=mats[i], max_edges=3)

    dt2, _, _, _ = side_frames(ctx, torch, dist, world, dev, submit_each, 2, warm=1)
    out["one_batch_per_object_ms_per_frame"] = round(dt2 / 2 * 1e3, 3)
    return out


def c3_side(torch, dist, dev, local_rank, tris, q2_tris, frames=24):
    """the 4K workload beside the headline (one GPU): configuration 3, and its variant with
    differing vertex lights (filled in draw order)"""
    from raytracing_amd import gpu as G
    out = {}
    for name, n, nf in (("c3", tris, frames), ("c3-q2", q2_tris, 8)):
        sc = make_scene(name, n)
        ctx = G.Context(sc.width, sc.height, device=local_rank)
        for h, (tex, lg) in enumerate(sc.textures):
            ctx.texture_upload(h, tex, lg)
        d_verts, d_polys, d_pv, d_aux = (torch.from_numpy(a).to(dev) for a in (sc.verts, sc.polys, sc.polyverts, sc.polyaux))

        def submit(k):
            ctx.clear(0)
            ctx.submit_object_dev(d_verts, d_polys, d_pv, sc.render_type, sc.log_focus, d_polyaux=d_aux, matrix=sc.matrix,
                                  force_linear=sc.force_linear, max_edges=3)

        # (four warm-up frames: the context has four sets of work arrays, each allocated by the first batch that uses it)
        dt, frags, n_ctr, err = side_frames(ctx, torch, dist, 1, dev, submit, nf, warm=4)
        out[name.replace("-", "_")] = {
            "workload": f"{name}: {sc.n_polys} textured triangles at {sc.width}x{sc.height}, log_focus {sc.log_focus}, "
                        + ("one light value per triangle" if name == "c3" else "differing vertex lights (draw-order fill)"),
            "frames": nf, "ms_per_frame": round(dt / nf * 1e3, 4), "mtris_per_s": round(sc.n_polys * nf / dt / 1e6, 3),
            "mfragments_per_s": round(frags / dt / 1e6, 3), "unsupported_polys": n_ctr["unsupported_polys"]}
        if err:
            out[name.replace("-", "_")]["flush_report"] = err[:200]
        ctx.close()
        del d_verts, d_polys, d_pv, d_aux, sc
    return out


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """--gpus N > 1 without a launcher: start N ranks (one per GPU) through torch.distributed.run.
    Nothing in this process has touched a GPU; the children's output passes straight through."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def default_queue_run(args):
    """The headline step once more in a child process that keeps HIP's default of four hardware
    queues (started and finished before this process touches the GPU) -> dict or None"""
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(min(args.steps, 60)), "--warmup", str(args.warmup),
           "--workload", args.workload, "--tris", str(args.tris), "--hw-queues", "4", "--headline-only"]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        j = json.loads(line)
        return {"hw_queues": 4, "steps": j["steps"], "ms_per_step": j["ms_per_step"], "value": j["value"], "unit": j["unit"]}
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2-small")
    ap.add_argument("--tris", type=int, default=0, help="triangles per frame (0: the configuration's own: 1M, 10M for c3*)")
    ap.add_argument("--split", default="frames", choices=["frames", "bands"])
    ap.add_argument("--route", action="store_true",
                    help="with --split bands: every rank works out the rows of its share of the polygons, the ranks exchange "
                         "polygon indices by band, and each sets up only the polygons that can touch its band")
    ap.add_argument("--no-pipelining", action="store_true", help="issue every kernel on one stream")
    ap.add_argument("--host-arrays", action="store_true",
                    help="submit from host memory (the boundary's host-pointer entry point): PCIe-inclusive rate, "
                         "reported for reference, never the headline value")
    ap.add_argument("--cpu-cores", type=int, default=min(16, os.cpu_count() or 1),
                    help="processes of the all-cores CPU baseline (0 = skip it)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo to rehearse on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses device 0")
    ap.add_argument("--tile-width", type=int, default=0, choices=[0, 64, 128], help="fill tile width (0: chosen by the library)")
    ap.add_argument("--no-frame-overlap", action="store_true", help="keep every frame in one buffer pair (no two frames in flight)")
    ap.add_argument("--no-alone-pass", action="store_true", help="skip the one-stream frames that time the slowest kernel alone")
    ap.add_argument("--no-large-class", action="store_true", help="skip the C2-large frames after the headline measurement")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg and the frame check")
    ap.add_argument("--no-side-lines", action="store_true", help="skip the c4 / c3_4k / default_hw_queues measurements")
    ap.add_argument("--headline-only", action="store_true", help="the timed region and nothing else")
    ap.add_argument("--sample-every", type=int, default=0,
                    help="time the slowest kernel inside the region on every Nth frame (0: every 4th, or every frame of a short region)")
    ap.add_argument("--hw-queues", type=int, default=8,
                    help="GPU_MAX_HW_QUEUES for this process (HIP's default is 4; the library keeps four streams busy); "
                         "0 leaves the environment as it is (the library then sets 8 if the variable is unset)")
    ap.add_argument("--wake-frames", type=int, default=100,
                    help="untimed frames drawn before the probe and warm-up frames, so that the device is out of its idle power state "
                         "when the warm-up starts (the CPU legs before it leave the GPU idle for seconds; measured on 20-frame regions: "
                         "0.120 ms per frame without, 0.117 with 100; reported as config.device_wake_frames)")
    ap.add_argument("--hip-first", action="store_true",
                    help="start HIP before the library is loaded (with --hw-queues 0 and the variable unset: the process keeps HIP's four queues)")
    args = ap.parse_args()
    if args.headline_only:
        args.no_alone_pass = args.no_large_class = args.no_cpu = args.no_side_lines = True
    if args.tris <= 0:
        args.tris = 10_000_000 if args.workload.startswith("c3") else 1_000_000

    # HIP gives a process GPU_MAX_HW_QUEUES hardware queues (4 unless set) and lets streams share
    # them; the library keeps two frames in flight on four streams of its own beside the caller's,
    # and two of them sharing one queue run one after the other.  Read when HIP starts: set first.
    if not 0 <= args.hw_queues <= 32:
        raise SystemExit("--hw-queues outside 0..32")
    if args.hw_queues:
        os.environ["GPU_MAX_HW_QUEUES"] = str(args.hw_queues)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU")

    # child processes start before this process touches the GPU: the all-cores CPU baseline and
    # the default-queue run
    all_cores = dflt_q = None
    headline = args.workload == "c2-small" and not args.host_arrays and args.split == "frames" and not args.no_pipelining
    if rank == 0 and world == 1 and not args.no_cpu and args.cpu_cores > 1 and args.workload.startswith("c2"):
        all_cores = cpu_all_cores(args.workload, args.tris, args.cpu_cores)
    if rank == 0 and world == 1 and headline and not args.no_side_lines and args.hw_queues != 4:
        dflt_q = default_queue_run(args)

    import torch
    import torch.distributed as dist
    if args.hip_first:
        torch.cuda.init()
    from raytracing_amd import bands as B, gpu as G

    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
        observed_world = dist.get_world_size()
        if observed_world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {observed_world} ranks")
    else:
        observed_world = 1

    if args.workload == "shade":
        bench_shade(args, rank, local_rank, world)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    sc = make_scene(args.workload, args.tris)
    dev = torch.device("cuda", local_rank)
    d_verts = torch.from_numpy(sc.verts).to(dev)
    d_polys = torch.from_numpy(sc.polys).to(dev)
    d_pv = torch.from_numpy(sc.polyverts).to(dev)
    d_aux = torch.from_numpy(sc.polyaux).to(dev) if sc.polyaux is not None else None

    ctx = G.Context(sc.width, sc.height, device=local_rank)
    for h, (tex, lg) in enumerate(sc.textures):
        ctx.texture_upload(h, tex, lg)
    if args.no_pipelining:
        ctx.set_pipelining(False)
    if args.tile_width:
        ctx.set_tile_width(args.tile_width)
    if args.no_frame_overlap:
        ctx.set_frame_overlap(False)

    bands = args.split == "bands"
    if bands:
        # the frame lives in torch tensors so that the gather can address it
        f_c = torch.empty((sc.height, sc.width), dtype=torch.int32, device=dev)
        f_z = torch.empty((sc.height, sc.width), dtype=torch.int32, device=dev)
        ctx.bind_framebuffer(f_c, f_z)
        y0, y1 = B.band_rows(rank, world, sc.height)
        if y0 < y1:
            ctx.set_band(y0, y1)

    # configuration 4: this rank's frames of the 1024-frame animated batch, each under its own
    # matrix (T_set_self_rotation + T_cancatinate_self_world are a few dozen host multiplies)
    animated = args.workload == "c4"
    if animated:
        from raytracing_amd import frames as F
        my_frames = list(F.frames_for_rank(1024, rank, world))
        mats = [c4_matrix(f) for f in my_frames]
    frame_no = [0]
    routed = [0]

    def step():
        m = sc.matrix
        if animated:
            m = mats[frame_no[0] % len(mats)]
            frame_no[0] += 1
        if args.host_arrays:
            ctx.clear(0)
            ctx.submit_object(sc.verts, sc.polys, sc.polyverts, sc.render_type, sc.log_focus, polyaux=sc.polyaux,
                              matrix=m, force_linear=sc.force_linear, max_edges=3)
        elif bands and args.route:
            # routing is part of the step: rows of this rank's share, lists by band, the exchange
            lo, hi = B.share(rank, world, sc.n_polys)
            rows = ctx.polygon_rows_dev(d_verts, d_polys[lo:hi], d_pv, sc.log_focus, matrix=m, max_edges=3)
            mine = B.exchange_lists(B.lists_by_band(rows, lo, world, sc.height), rank, world).long()
            routed[0] += int(mine.numel())
            sel_polys = d_polys[mine].contiguous()
            sel_aux = d_aux[mine].contiguous() if d_aux is not None else None
            if y0 < y1:
                ctx.clear(0)
                if mine.numel():
                    ctx.submit_object_dev(d_verts, sel_polys, d_pv, sc.render_type, sc.log_focus, d_polyaux=sel_aux,
                                          matrix=m, force_linear=sc.force_linear, max_edges=3)
            ctx.flush()                          # the selected arrays are read by the library's own streams: keep them until then
        elif not bands or y0 < y1:
            ctx.clear(0)
            ctx.submit_object_dev(d_verts, d_polys, d_pv, sc.render_type, sc.log_focus, d_polyaux=d_aux,
                                  matrix=m, force_linear=sc.force_linear, max_edges=3)
        if bands and world > 1:
            ctx.flush()                          # the band is complete before it is sent
            B.gather_bands([f_c, f_z], rank, world, sc.height)

    # Which stage is the slowest is found on a few frames on one stream, every stage timed: with
    # two frames in flight the stages stretch each other and the order of their durations changes
    # from run to run.  The warm-up frames proper follow, and the timed region then carries events
    # only around that stage's kernel (events cost stream time: seven per frame were a tenth of it).
    probe = not args.no_pipelining and not bands and not args.host_arrays
    wake = args.wake_frames if probe else 0
    for _ in range(wake):
        step()
    if wake:
        ctx.flush()
        frame_no[0] = 0
    ctx.set_profiling(True)
    if probe:
        ctx.set_pipelining(False)
        for i in range(4):
            step()
            if i == 0:
                ctx.stage_ms()                   # the first frame allocates and loads code: not representative
        warm_stages, warm_submits = ctx.stage_ms()
        ctx.set_pipelining(True)
    for i in range(args.warmup):
        step()
        if i == 0 and args.warmup > 1:
            ctx.stage_ms()
    if not probe:
        warm_stages, warm_submits = ctx.stage_ms()
    dom = max(warm_stages, key=lambda k: warm_stages[k]) if warm_submits else "fill"
    ctx.reset_counters()
    frame_no[0] = 0
    # ... and only around every fourth launch of it: an event holds its stream for a few
    # microseconds, which two-stream frames of under 0.2 ms feel
    sample_every = args.sample_every if args.sample_every > 0 else (4 if args.steps >= 8 else 1)
    if args.sample_every < 0:
        ctx.set_profiling(False)
    else:
        ctx.set_profiling(dom, every=sample_every)
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
    # The same kernel with the chip to itself: a few frames on one stream right after the timed
    # region, HIP events on the kernel's own dispatch.  (In the region two frames are in flight:
    # two launches of the kernel share the chip and each lasts longer than a step.)
    alone_ms = None
    if rank == 0 and not args.no_pipelining and not bands and not args.host_arrays and not args.no_alone_pass and not animated:
        ctx.set_pipelining(False)
        ctx.set_profiling(dom, every=1)
        for _ in range(10):
            step()
        a_stages, a_submits = ctx.stage_ms()
        alone_ms = a_stages[dom] / max(a_submits, 1)
        ctx.set_pipelining(True)
        counters_after = ctx.counters()
        assert counters_after["fragments"] == counters["fragments"] // max(args.steps, 1) * (args.steps + 10)
    ctx.set_profiling(False)
    frags = torch.tensor([counters["fragments"]], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(frags)                  # bands: the frame's total; frames: all ranks' frames
    frags = int(frags.item())

    # side measurements on every rank (they synchronise the ranks like the headline region)
    side = {}
    if headline and not args.no_large_class:
        side["large_class"] = large_class(ctx, torch, dev, args.tris)
    if headline and not args.no_side_lines:
        side["c4"] = c4_side(ctx, torch, dist, rank, world, dev, d_verts, d_polys, d_pv, sc)
        side["many_objects"] = many_objects_side(ctx, torch, dist, world, dev, sc, d_verts, d_polys, d_pv)

    if rank == 0:
        total_tris = sc.n_polys * args.steps * (1 if bands else world)
        frame_bytes = BYTES_PER_TRI * sc.n_polys + BYTES_PER_PIXEL * sc.width * sc.height
        step_ms = dt_max / args.steps * 1e3
        in_region_ms = stages[dom] / max(submits, 1)
        # The kernel's own compulsory bytes per launch.  Set-up reads the inputs (160 B a
        # triangle) once and writes a record per polygon that reaches the fill; the fill reads
        # each such record once (64 / 160 / 208 B: flat / Gouraud / textured), one 4-byte list
        # entry per (polygon, tile) pair, and writes the frame once.
        per_frame = max(args.steps, 1) * (1 if bands else 1)
        drawn = counters["polys_drawn"] // per_frame
        refs = counters["tile_refs"] // per_frame
        rec_bytes = {2: 64, 4: 160, 8: 208}[sc.render_type]
        own_bytes = {"fill": BYTES_PER_PIXEL * sc.width * sc.height + rec_bytes * drawn + 4 * refs,
                     "setup": BYTES_PER_TRI * sc.n_polys + (rec_bytes + 16) * drawn,
                     "bin": 16 * sc.n_polys + 4 * refs, "scan": 16 * sc.n_polys}.get(dom, frame_bytes)
        kernel_ms = alone_ms if alone_ms else in_region_ms
        achieved = own_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        traffic, traffic_note = (None, None) if bands else measured_traffic(args.workload, sc.n_polys, dom)
        what = "Gouraud Z-buffered" if sc.render_type == 4 else ("textured" if sc.render_type == 8 else "flat")
        out = {
            "metric": "Z-tested shaded Mfragments/s + Mtris/s at 1080p/4K, 1/2/4/8 MI355X",
            "value": round(total_tris / dt_max / 1e6, 3),
            "unit": "Mtris/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(step_ms, 4),
            "higher_is_better": True,
            "scaling": "strong" if bands else "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic" + (", submitted from host memory every frame (PCIe-inclusive)" if args.host_arrays else ""),
            "config": {"workload": f"{args.workload}: {sc.n_polys} {what} triangles per frame at {sc.width}x{sc.height}, one frame per "
                                   f"step per GPU" + (", per-frame object matrix (frames rank, rank + world, ... of 1024)" if animated else ""),
                       "parallelism": (f"sort-first row bands x{world}, bands gathered to rank 0"
                                       + (", polygons routed to their bands before set-up" if args.route else "") if bands
                                       else f"frame-parallel x{world}"), "tile": "64x32 or 128x32, chosen by the library",
                       "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "world_size_observed": observed_world,
                       "device_wake_frames": wake,
                       "backend": (args.backend if world > 1 else None)},
            "mfragments_per_s": round(frags / dt_max / 1e6, 3),
            "fragments_per_frame": frags // max(args.steps, 1) // (1 if bands else world),
            "stage_ms_per_frame": {k: round(v / max(warm_submits, 1), 4) for k, v in warm_stages.items()},
            "stage_ms_measured_on": (f"{warm_submits} frame(s) on one stream before the warm-up, all stages timed" if probe else
                                     f"{warm_submits} warm-up frame(s), all stages timed"),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "traffic_note": traffic_note,
                         "kernel_ms": round(kernel_ms, 4),
                         "kernel_ms_measured_on": ("10 frames on one stream right after the timed region, HIP events on the kernel's own "
                                                   "dispatch (the kernel with the chip to itself)" if alone_ms else
                                                   f"{submits} of the {args.steps} measured frames (every {sample_every}th)"),
                         "algorithmic_bytes_per_launch": own_bytes,
                         "algorithmic_bytes_are": "this kernel's compulsory traffic: the frame written once, one record per polygon that "
                                                  "reaches the fill, one list entry per (polygon, tile) pair" if dom == "fill" else
                                                  "this kernel's compulsory traffic",
                         "path_bytes_per_step": frame_bytes,
                         "whole_step_frac": round(frame_bytes / (dt_max / args.steps) / 1e9 / HBM_PEAK_GBS, 5),
                         "in_region": {"kernel_ms": round(in_region_ms, 4), "launches_timed": submits,
                                       "note": "two frames are in flight in the timed region: two launches of the kernel share the "
                                               "chip, so one launch lasts longer than a step"}},
        }
        # (the committed counter pass is of whole frames on one GPU: no such line for a band of a frame)
        vr = None if bands else valu_roofline(args.workload, sc.n_polys, dom, kernel_ms)
        if vr:
            out["roofline_valu"] = vr
        if dflt_q:
            out["default_hw_queues"] = dflt_q
        if bands and args.route:
            out["routed_polys_per_frame_rank0"] = routed[0] // max(args.steps + args.warmup, 1)
        out.update(side)
        print_line = out
    else:
        print_line = None

    # one GPU: the 4K workload and the CPU legs beside the headline
    if rank == 0 and world == 1:
        if not args.no_cpu:
            base, oc, oz = cpu_baseline(sc, passes=2 if sc.n_polys <= 2_000_000 else 1)
            # the frame the CPU leg drew (the side measurements have drawn other frames since)
            ctx.clear(0)
            ctx.submit_object_dev(d_verts, d_polys, d_pv, sc.render_type, sc.log_focus, d_polyaux=d_aux, matrix=sc.matrix,
                                  force_linear=sc.force_linear, max_edges=3)
            c, z = ctx.readback(check=False)
            print_line["cpu_baseline"] = base
            print_line["cpu_baseline_touched_rows"] = cpu_touched_rows(sc)
            if all_cores:
                print_line["cpu_baseline_all_cores"] = all_cores
            print_line["frame_matches_cpu"] = bool(np.array_equal(c, oc) and np.array_equal(z, oz))
        if headline and not args.no_side_lines:
            ctx.close()
            del d_verts, d_polys, d_pv
            print_line["c3_4k"] = c3_side(torch, dist, dev, local_rank, 10_000_000, 10_000_000)
    if rank == 0:
        assert print_line["n_gpus"] == args.gpus
        print(json.dumps(print_line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
