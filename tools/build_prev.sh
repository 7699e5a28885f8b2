#!/bin/bash
# Build the committed HEAD sources as mujoco_warp_amd/libmjhip_prev.so for same-session A/B (tools/ab.sh MJH_LIB=...)
set -e
rm -rf build/prev; mkdir -p build/prev/pkg/csrc build/prev/include
for f in mjhip.hip dev_common.hpp smooth.hpp collide.hpp constraint.hpp solver.hpp integrate.hpp; do git show ${1:-HEAD}:mujoco_warp_amd/csrc/$f > build/prev/pkg/csrc/$f; done
git show ${1:-HEAD}:include/mjhip.h > build/prev/include/mjhip.h
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -o mujoco_warp_amd/libmjhip_prev.so build/prev/pkg/csrc/mjhip.hip
