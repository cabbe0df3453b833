#!/bin/bash
# Diagnostic builds of the blend kernel (rasterizer.hip -DAMAV_ABLATE=n) timed on the bench workload, plus the
# residency sweep (AMAV_RENDER_WAVES) and the per-phase clock stamps.  Run through gpurun from the repo root:
#   tools/ablate_render.sh [tag]       -> gpurun_out/ablate_<tag>.txt
tag=${1:-x}
out=gpurun_out/ablate_${tag}.txt
mkdir -p gpurun_out /tmp/amav_ablate
cd audio-motion-avatar_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize"
for n in 3 5 7; do
  /opt/rocm/bin/hipcc $FLAGS -DAMAV_ABLATE=$n -c rasterizer.hip -o /tmp/amav_ablate/rasterizer_$n.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC api.o /tmp/amav_ablate/rasterizer_$n.o lbs.o triplane.o attention.o frames.o splat.o cloud.o gemm.o -L/opt/rocm/lib -lhipblaslt -o /tmp/amav_ablate/libamav_$n.so || exit 1
done
cd ../..
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-path --no-refiner --no-extra-configs"
get() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms blend, %.4f ms step' % (d['roofline']['avg_launch_ms'], d['step_device_ms']['median']))"; }
{
echo "product build:        $($B 2>/dev/null | get)"
echo "ABLATE=3 (no blend):  $(AMAV_LIB=/tmp/amav_ablate/libamav_3.so $B 2>/dev/null | get)"
echo "ABLATE=5 (no bg):     $(AMAV_LIB=/tmp/amav_ablate/libamav_5.so $B 2>/dev/null | get)"
echo "ABLATE=7 (neither):   $(AMAV_LIB=/tmp/amav_ablate/libamav_7.so $B 2>/dev/null | get)"
for w in 4 8 12 16 20; do
  echo "AMAV_RENDER_WAVES=$w: $(AMAV_RENDER_WAVES=$w $B 2>/dev/null | get)"
done
python tools/stamp_render.py
} > $out 2>&1
cat $out
