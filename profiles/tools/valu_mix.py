#!/usr/bin/env python3
"""Static instruction mix of the tile fill: compiles raytracing_amd/csrc/lt_kernels.hip to gfx950 assembly
(needs hipcc; no GPU) and counts, over the hot blocks of raster_kernel<4, 64, 256> -- the triangle row step of
the depth pass and the colour pass -- which share of the integer instructions are the simple two-operand
forms that a SIMD issues in about two cycles (add, sub, shifts, and / or / xor, compares to vcc, select
on vcc) as against about four for everything else (profiles/tools/issue_rate.hip measures both).  The
SQ_INSTS_VALU_INT32 counter does not tell the two apart; bench.py splits it by this share.

  python profiles/tools/valu_mix.py            -> profiles/r3_valu_mix.json
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "raytracing_amd", "csrc")
SIMPLE = re.compile(r"^v_(add_u32|sub_u32|subrev_u32|add_co_u32|sub_co_u32|lshlrev_b32|lshrrev_b32|ashrrev_i32|and_b32|or_b32|xor_b32|"
                    r"not_b32|mov_b32|cndmask_b32|cmp_[a-z]+_[iu]32)_e32$")


def source_sha():
    h = hashlib.sha256()
    for f in ("lt_kernels.hip", "lt_api.hip", "lt_kernels.h", "lt_math.h"):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def kernel_asm(asm, mangled):
    out, on = [], False
    for line in asm.splitlines():
        if line.startswith(mangled + ":"):
            on = True
        if on:
            out.append(line)
            if "s_endpgm" in line:
                break
    return out


def main():
    with tempfile.TemporaryDirectory() as d:
        s = os.path.join(d, "k.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                               os.path.join(CSRC, "lt_kernels.hip"), "-o", s], stderr=subprocess.DEVNULL)
        asm = open(s).read()
    lines = kernel_asm(asm, "_ZN2lt13raster_kernelILi4ELi64ELi256EEEvNS_10RasterArgsE")
    ops = [l.split()[0] for l in lines if re.match(r"^\s+v_", l)]
    integer = [o for o in ops if not re.search(r"_f(16|32|64)|cvt|rcp|mfma|mov_b64", o)]
    simple = [o for o in integer if SIMPLE.match(o)]
    hist = {}
    for o in ops:
        hist[o] = hist.get(o, 0) + 1
    out = {"source_sha256": source_sha(),
           "kernels": {"fill": {"int32_simple_share": round(len(simple) / max(len(integer), 1), 4), "valu_static": len(ops),
                                "integer_static": len(integer), "simple_static": len(simple)}},
           "histogram_raster_kernel_4_64": dict(sorted(hist.items(), key=lambda kv: -kv[1]))}
    path = os.path.join(ROOT, "profiles", "r3_valu_mix.json")
    json.dump(out, open(path, "w"), indent=1)
    print(path, out["kernels"])


if __name__ == "__main__":
    sys.exit(main())
