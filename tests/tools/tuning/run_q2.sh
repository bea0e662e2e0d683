#!/bin/bash
for rep in 1 2; do
for v in "$@"; do
  cp scratch/libltgpu_$v.so raytracing_amd/libltgpu.so || exit 1
  timeout -k 10 150 python bench.py --workload c3-q2 --steps 8 --warmup 2 --headline-only > gpurun_out/var_${v}_q2.log 2>&1 || { echo "FAILED $v"; tail -5 gpurun_out/var_${v}_q2.log; exit 1; }
  python - <<PY
import json
l=[x for x in open("gpurun_out/var_${v}_q2.log") if x.startswith("{")][-1]
d=json.loads(l); print("$v rep$rep c3-q2 ms_per_step", d["ms_per_step"], "frags/frame", d["fragments_per_frame"])
PY
done
done
cp scratch/libltgpu_cur.so raytracing_amd/libltgpu.so
