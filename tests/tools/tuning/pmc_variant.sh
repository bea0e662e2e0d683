#!/bin/bash
# usage: pmc_variant.sh v1 v2 ...  : SQ counters of the fill per launch for each library variant (one stream)
export TMPDIR=/tmp
for v in "$@"; do
  cp scratch/libltgpu_$v.so raytracing_amd/libltgpu.so || exit 1
  rm -rf gpurun_out/pmc_$v
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_ANY -d gpurun_out/pmc_$v -o p --output-format csv -- python3 bench.py --headline-only --steps 10 --warmup 3 --no-pipelining > gpurun_out/pmc_$v.log 2>&1 || { echo FAILED $v; tail -5 gpurun_out/pmc_$v.log; exit 1; }
  python3 profiles/pmc_summary.py gpurun_out/pmc_$v 2>&1 | grep -A10 "raster_kernel" | head -14
  find gpurun_out/pmc_$v -name '*.csv' -size +2M -delete
done
cp scratch/libltgpu_cur.so raytracing_amd/libltgpu.so
