#!/bin/bash
# A/B of two builds of the library on the same box, alternating runs (box-to-box spread is larger than most changes):
#   build_ab/libamav_head.so vs build_ab/libamav_new.so, the render bench line of each, three times.  Through gpurun.
B="python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-full-path --no-refiner --no-extra-configs"
get() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms blend, %.4f ms step, %.0f frames/s' % (d['roofline']['avg_launch_ms'], d['step_device_ms']['median'], d['value']))"; }
for i in 1 2 3; do
echo "new : $(AMAV_LIB=$GRAFT_REPO_ROOT/build_ab/libamav_new.so $B 2>/dev/null | get)"
echo "HEAD: $(AMAV_LIB=$GRAFT_REPO_ROOT/build_ab/libamav_head.so $B 2>/dev/null | get)"
done
