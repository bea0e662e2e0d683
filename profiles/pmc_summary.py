#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel launch."""
import collections
import csv
import glob
import sys


def main(d):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if "rocclr" in k:
            continue
        print(k[:70])
        for c, v in sorted(cs.items()):
            print(f"   {c:28s} n={len(v):3d} mean={sum(v) / len(v):16.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
