#!/bin/bash
# builds ablation variants of the cell kernels into aspect_amd/csrc/abl<n>/libaspect_amd.so (load with AGMG_LIB=...); n is a bit mask:
# 1 no slot stores, 2 no src gather, 4 no index loads, 8 no per-q viscosity loads,
# 16 no per-line sched_barrier, 64 no cell arithmetic (block kernel), 128 no node sums (block kernel), 32 launch bound 3 waves/SIMD, 256 no XCD-aware block mapping, 1024 phase stamps (block kernel),
# 4096 phase stamps (quad brick kernel)
set -e
cd "$(dirname "$0")/../aspect_amd/csrc"
for n in "$@"; do
  d=abl$n; mkdir -p $d
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DAGMG_ABLATE=$n -c kernels_ops.hip -o $d/kernels_ops.o &
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DAGMG_ABLATE=$n -c kernels_brick.hip -o $d/kernels_brick.o &
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DAGMG_ABLATE=$n -c kernels_brickq.hip -o $d/kernels_brickq.o &
  wait
  hipcc --offload-arch=gfx950 -shared -o $d/libaspect_amd.so agmg.o $d/kernels_ops.o $d/kernels_brick.o $d/kernels_brickq.o kernels_transfer.o kernels_vec.o scratch.o transport_rccl.o -ldl
done
