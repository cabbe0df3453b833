#!/bin/bash
# Diagnostic build of the binning kernel (-DAMAV_BIN_STAMPS) + tools/stamp_bin.py: where a frame's block spends its time.
# Run through gpurun.
mkdir -p /tmp/amav_bin && cd audio-motion-avatar_amd/csrc || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DAMAV_BIN_STAMPS ${AMAV_BIN_FLAGS} -c rasterizer.hip -o /tmp/amav_bin/rasterizer.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC api.o /tmp/amav_bin/rasterizer.o lbs.o triplane.o attention.o frames.o splat.o cloud.o gemm.o -L/opt/rocm/lib -lhipblaslt -o /tmp/amav_bin/libamav.so || exit 1
cd ../..
AMAV_LIB=/tmp/amav_bin/libamav.so python tools/stamp_bin.py "${1:-250}"
