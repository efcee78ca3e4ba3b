#!/bin/bash
# Where a workgroup of k3_merge (level 3 of the bucket sort) spends its time: builds the library with -DPANDA_K3_STAMPS (msm_sort.hip:
# s_memrealtime at the phase boundaries of thread 0 of every 16th workgroup) into tools/bin/libpanda-k3stamps.so -- run HERE, before
# gpurun -- and prints the phases of a 2^24-point call:  on the GPU box:  python3 tools/k3_stamps.py [log_n] [wide-merge mode 0|1|2]
set -e
cd "$(dirname "$0")/../panda_amd/csrc"
mkdir -p ../../tools/bin
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DPANDA_K3_STAMPS -c msm_sort.hip -o /tmp/msm_sort_k3stamps.o
objs=""
for f in shim msm msm_bn254 msm_bls377 msm_bls381 msm_bn254g2 host_msm ntt multi_gpu debug_gen; do objs="$objs $f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/bin/libpanda-k3stamps.so $objs /tmp/msm_sort_k3stamps.o -L/opt/rocm/lib -lrccl
echo built tools/bin/libpanda-k3stamps.so
