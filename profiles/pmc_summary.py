#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc runs: mean counter value per kernel launch.

  pmc_summary.py DIR [DIR ...]            print per-kernel means of every counter found
  pmc_summary.py --json OUT DIR [DIR ...] also write them as JSON (kernel -> counter -> mean per launch; "_meta" holds the
                                          sha256 of the kernel sources and, if profiles/r3_valu_mix.json is of the same
                                          sources, the static instruction mix of the fill)

Each DIR is the output directory of one `rocprofv3 --pmc ...` pass (FETCH_SIZE and WRITE_SIZE
need separate passes).  FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them; on
gfx950 FETCH_SIZE counts 16-byte-per-lane streaming reads at half their bytes.
"""
import collections
import csv
import glob
import json
import os
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


def source_sha():
    """sha256 over the sources of the HIP library: bench.py quotes these counters only for the code they were taken on"""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("lt_kernels.hip", "lt_api.hip", "lt_kernels.h", "lt_math.h"):
        with open(os.path.join(root, "raytracing_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


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
        meta = {"source_sha256": source_sha()}
        mix = os.path.join(os.path.dirname(os.path.abspath(__file__)), "r3_valu_mix.json")
        if os.path.exists(mix):
            m = json.load(open(mix))
            if m.get("source_sha256") == meta["source_sha256"]:
                meta["valu_mix"] = m.get("kernels", {})
        merged["_meta"] = meta
        json.dump(merged, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1:])
