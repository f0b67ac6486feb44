#!/bin/bash
# builds ablation variants of the cell kernel into build/abl<n>/libaspect_amd.so; n is a bit mask:
# 1 no slot stores, 2 no src gather, 4 no index loads, 8 no per-q viscosity loads,
# 16 no per-line sched_barrier, 32 launch bound 3 waves/SIMD, 256 no XCD-aware block mapping, 1024 phase stamps
set -e
cd "$(dirname "$0")/../aspect_amd/csrc"
for n in "$@"; do
  d=../../build/abl$n; mkdir -p $d
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DAGMG_ABLATE=$n -c kernels_ops.hip -o $d/kernels_ops.o
  hipcc --offload-arch=gfx950 -shared -o $d/libaspect_amd.so agmg.o $d/kernels_ops.o kernels_transfer.o kernels_vec.o transport_rccl.o -ldl
done
