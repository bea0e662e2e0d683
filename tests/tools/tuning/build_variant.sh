#!/bin/bash
# usage: build_variant.sh NAME 'sed-expr-for-lt_kernels.h' 'sed-expr-for-lt_kernels.hip'
set -e
N=$1
D=/root/repo/scratch/src_$N
rm -rf $D && mkdir -p $D
cp /root/repo/raytracing_amd/csrc/* $D/
[ -n "$2" ] && sed -i -E "$2" $D/lt_kernels.h
[ -n "$3" ] && sed -i -E "$3" $D/lt_kernels.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $D/lt_kernels.hip $D/lt_api.hip -o /root/repo/scratch/libltgpu_$N.so
echo built $N
