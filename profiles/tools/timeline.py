#!/usr/bin/env python3
"""Prints the steady state of a time-stamped kernel trace (profiles/tools/trace_timeline.sh) kernel by kernel:
start, end and duration in microseconds relative to one fill's start, hardware queue and stream.

  python profiles/tools/timeline.py [gpurun_out/trace_pipe/kernel_trace.csv]
"""
import csv
import sys

SHORT = (("raster", "FILL"), ("setup_fast", "fast"), ("setup_queues", "queue"), ("count_kernel", "count"), ("scan_", "scan"),
         ("scatter", "scat"), ("fillBuffer", "memset"), ("copyBuffer", "copy"))


def short(name):
    for k, v in SHORT:
        if k in name:
            return v
    return name[:12]


def main(path):
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"], r["Stream_Id"]) for r in rows)
    fills = [e for e in ev if e[2] == "FILL"]
    t0 = fills[-30][0] if len(fills) > 30 else fills[0][0]
    for s, e, n, q, st in ev:
        if t0 - 100000 <= s < t0 + 500000:
            print("%8.1f %8.1f %6.1f  q%s s%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, st, n))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/trace_pipe/kernel_trace.csv")
