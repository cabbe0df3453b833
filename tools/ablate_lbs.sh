#!/bin/bash
# Diagnostic builds of the skinning kernel (-DAMAV_LBS_ABLATE=1: no epilogue, =2: one table chunk only) timed stand-alone
# with tools/bench_lbs.py (nothing consumes the vertices).  Run through gpurun.
mkdir -p /tmp/amav_lbs && cd audio-motion-avatar_amd/csrc || exit 1
for a in 0 1 2; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DAMAV_LBS_ABLATE=$a -c lbs.hip -o /tmp/amav_lbs/lbs.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC api.o rasterizer.o /tmp/amav_lbs/lbs.o triplane.o attention.o frames.o splat.o cloud.o gemm.o -L/opt/rocm/lib -lhipblaslt -o /tmp/amav_lbs/libamav$a.so || exit 1
done
cd ../..
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in 0 1 2; do
  rm -rf gpurun_out/prof_lbs$a
  AMAV_LIB=/tmp/amav_lbs/libamav$a.so rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_lbs$a -- python tools/bench_lbs.py 250 > gpurun_out/prof_lbs$a.log 2>&1
  echo "AMAV_LBS_ABLATE=$a:"; python tools/kernel_stats.py gpurun_out/prof_lbs$a 4 | grep -i "lbs\|skin"
done
