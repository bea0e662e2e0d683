#!/bin/bash
# Time-stamped kernel trace of the pipelined headline run (gpurun_out/trace_pipe/kernel_trace.csv);
# python profiles/tools/timeline.py prints the steady state stream by stream.
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/trace_pipe
rm -rf $O && mkdir -p $O
timeout -k 10 170 rocprofv3 --kernel-trace -d $O/t -o t --output-format csv -- python3 bench.py --headline-only --steps 40 --warmup 3 > $O/t.log 2>&1 || { echo trace failed; tail -3 $O/t.log; exit 1; }
cp $(find $O/t -name '*kernel_trace.csv' | head -1) $O/kernel_trace.csv; rm -rf $O/t
wc -l $O/kernel_trace.csv; head -2 $O/kernel_trace.csv
