#!/bin/bash
# Kernel trace and FETCH_SIZE / WRITE_SIZE passes of --workload c3 on one stream (profiles/r3_kernel_stats_c3_4k_one_stream.csv,
# profiles/r3_pmc_c3_4k.json; copied by hand from gpurun_out/prof_c3).  Run on the GPU box from the repository root.
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/prof_c3
rm -rf $O && mkdir -p $O
A="--workload c3 --headline-only --steps 3 --warmup 1 --no-pipelining"
timeout -k 10 170 rocprofv3 --kernel-trace --stats -d $O/t -o t --output-format csv -- python3 bench.py $A > $O/t.log 2>&1 || { echo trace failed; tail -3 $O/t.log; exit 1; }
cp $(find $O/t -name '*kernel_stats.csv' | head -1) $O/kernel_stats_c3_one_stream.csv; rm -rf $O/t
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE -d $O/p_fetch -o p --output-format csv -- python3 bench.py $A > $O/pf.log 2>&1 || { echo fetch failed; exit 1; }
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE -d $O/p_write -o p --output-format csv -- python3 bench.py $A > $O/pw.log 2>&1 || { echo write failed; exit 1; }
python3 profiles/pmc_summary.py --json $O/pmc_c3_4k.json $O/p_fetch $O/p_write > $O/pmc_c3_4k.txt
find $O -name '*counter_collection.csv' -delete
find $O -type d -empty -delete
head -12 $O/kernel_stats_c3_one_stream.csv | cut -c1-160
cat $O/pmc_c3_4k.txt
