#!/bin/bash
# Diagnostic build of the blend kernel (-DAMAV_STAMP_DETAIL) + tools/stamp_render.py --detail: where a wave's time goes
# inside a tile (blend loops, the next tile's sort, background stores, everything else).  Run through gpurun.
mkdir -p /tmp/amav_detail && cd audio-motion-avatar_amd/csrc || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DAMAV_STAMP_DETAIL -c rasterizer.hip -o /tmp/amav_detail/rasterizer.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC api.o /tmp/amav_detail/rasterizer.o lbs.o triplane.o attention.o frames.o splat.o cloud.o gemm.o -L/opt/rocm/lib -lhipblaslt -o /tmp/amav_detail/libamav.so || exit 1
cd ../..
AMAV_LIB=/tmp/amav_detail/libamav.so python tools/stamp_render.py 250 --detail
