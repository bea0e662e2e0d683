#!/usr/bin/env python3
"""One worker of the all-cores CPU baseline (test infrastructure, started by bench.py
before the GPU is touched): renders the benchmark frame `passes` times with the reference
build (or the C restatement where that is not present) and prints the seconds it took.
The reference is not re-entrant, so cores are used by independent processes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    workload, tris, passes, start_at = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
    import bench
    sc = bench.make_scene(workload, tris)
    import oracle.port as P
    import oracle.ref as R
    vv = P.transform(sc.matrix, sc.verts)
    use_ref = os.path.exists(R.so_path(sc.width, sc.height)) and sc.render_type == 4
    eng = R.Ref(sc.width, sc.height) if use_ref else P.Oracle(sc.width, sc.height)
    if not use_ref:
        for h, (tex, lg) in enumerate(sc.textures):
            eng.set_texture(h, tex, lg)
    while time.time() < start_at:              # all workers start their timed loop together
        time.sleep(0.005)
    t0 = time.perf_counter()
    for _ in range(passes):
        eng.clear(0, 0, 0, 0)
        eng.render_polys(vv, sc.polys, sc.polyaux, sc.polyverts, sc.render_type, sc.log_focus, sc.force_linear)
    print(f"{time.perf_counter() - t0:.6f} {'reference' if use_ref else 'port'}", flush=True)


if __name__ == "__main__":
    main()
