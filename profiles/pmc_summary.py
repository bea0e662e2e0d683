#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc runs: mean counter value per kernel launch.

  pmc_summary.py DIR [DIR ...]            print per-kernel means of every counter found
  pmc_summary.py --json OUT DIR [DIR ...] also write them as JSON (kernel -> counter -> mean per launch)

Each DIR is the output directory of one `rocprofv3 --pmc ...` pass (FETCH_SIZE and WRITE_SIZE
need separate passes).  FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them; on
gfx950 FETCH_SIZE counts 16-byte-per-lane streaming reads at half their bytes.
"""
import collections
import csv
import glob
import json
import sys


def read(d):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    # one row per (dispatch, counter); sum over the rows of one dispatch first (per-XCD rows)
    per = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(f)):
        key = (r["Dispatch_Id"], r["Counter_Name"])
        per[key] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = r["Kernel_Name"]
    for (disp, ctr), v in per.items():
        agg[names[disp]][ctr].append(v)
    return agg


def main(argv):
    out = None
    if argv and argv[0] == "--json":
        out, argv = argv[1], argv[2:]
    merged = collections.defaultdict(dict)
    for d in argv:
        for k, cs in read(d).items():
            if "rocclr" in k:
                continue
            for c, v in cs.items():
                merged[k][c] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
    for k, cs in merged.items():
        print(k[:90])
        for c, e in sorted(cs.items()):
            print(f"   {c:28s} n={e['launches']:3d} mean={e['mean_per_launch']:16.1f}")
    if out:
        json.dump(merged, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1:])
