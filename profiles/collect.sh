#!/bin/bash
# Collects the kernel traces and counter passes that profiles/r3_* hold (python profiles/tools/valu_mix.py first, on the
# build machine: it needs hipcc).  Run on the GPU box from
# the repository root:  bash profiles/collect.sh   (results under gpurun_out/prof/, copied by hand).
# Counter passes are runs of their own (no trace options beside --pmc), each under its own time limit.
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8      # before the first HIP call of every process below (the profiler starts HIP before Python runs)
O=gpurun_out/prof
rm -rf $O && mkdir -p $O
trace() {   # name, bench arguments
  local n=$1; shift
  timeout -k 10 170 rocprofv3 --kernel-trace --stats -d $O/t_$n -o t --output-format csv -- python3 bench.py "$@" > $O/t_$n.log 2>&1 || { echo "trace $n failed"; tail -3 $O/t_$n.log; return 1; }
  cp $(find $O/t_$n -name '*kernel_stats.csv' | head -1) $O/kernel_stats_$n.csv
  rm -rf $O/t_$n
}
pmc() {     # name, pass, counters..., then -- bench arguments
  local n=$1 p=$2; shift 2
  local ctr=()
  while [ "$1" != "--" ]; do ctr+=("$1"); shift; done
  shift
  timeout -s KILL 170 rocprofv3 --pmc "${ctr[@]}" -d $O/p_${n}_$p -o p --output-format csv -- python3 bench.py "$@" > $O/p_${n}_$p.log 2>&1 || { echo "pmc $n $p failed"; tail -3 $O/p_${n}_$p.log; return 1; }
}
C2="--headline-only --steps 20 --warmup 3 --no-pipelining"
trace pipelined --headline-only --steps 40 --warmup 3 &&
trace one_stream $C2 &&
trace c3_4k --workload c3 --headline-only --steps 12 --warmup 2 &&
trace c3q2_4k --workload c3-q2 --headline-only --steps 6 --warmup 2 &&
trace c3q2_4k_one_stream --workload c3-q2 --headline-only --steps 6 --warmup 2 --no-pipelining &&
pmc c2small fetch FETCH_SIZE -- $C2 &&
pmc c2small write WRITE_SIZE -- $C2 &&
pmc c2small sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -- $C2 &&
pmc c2small sq2 SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES -- $C2 &&
pmc c2small sq3 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64 -- $C2 &&
pmc c2small sq4 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE -- $C2 &&
python3 profiles/pmc_summary.py --json $O/pmc_c2small.json $O/p_c2small_fetch $O/p_c2small_write $O/p_c2small_sq1 $O/p_c2small_sq2 $O/p_c2small_sq3 $O/p_c2small_sq4 > $O/pmc_c2small.txt &&
Q2="--workload c3-q2 --headline-only --steps 3 --warmup 1 --no-pipelining" &&
pmc c3q2 fetch FETCH_SIZE -- $Q2 &&
pmc c3q2 write WRITE_SIZE -- $Q2 &&
pmc c3q2 sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -- $Q2 &&
python3 profiles/pmc_summary.py --json $O/pmc_c3q2_4k.json $O/p_c3q2_fetch $O/p_c3q2_write $O/p_c3q2_sq1 > $O/pmc_c3q2_4k.txt
rc=$?
find $O -name '*counter_collection.csv' -delete
find $O -type d -empty -delete
ls $O
exit $rc
