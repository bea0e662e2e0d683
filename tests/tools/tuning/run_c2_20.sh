#!/bin/bash
# usage: run_c2_20.sh v1 v2 ...   (on the GPU box): the headline step in 20-frame regions (the round-end run's), five repeats
for rep in 1 2 3 4 5; do
for v in "$@"; do
  cp scratch/libltgpu_$v.so raytracing_amd/libltgpu.so || exit 1
  timeout -k 10 120 python bench.py --steps 20 --warmup 5 --headline-only > gpurun_out/var_${v}_c2.log 2>&1 || { echo "FAILED $v"; tail -5 gpurun_out/var_${v}_c2.log; exit 1; }
  python - <<PY
import json
l=[x for x in open("gpurun_out/var_${v}_c2.log") if x.startswith("{")][-1]
d=json.loads(l); print("$v rep$rep ms_per_step", d["ms_per_step"])
PY
done
done
cp scratch/libltgpu_cur.so raytracing_amd/libltgpu.so
