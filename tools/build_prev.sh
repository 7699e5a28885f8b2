#!/bin/bash
# Build the committed HEAD sources as mujoco_warp_amd/libmjhip_prev.so for same-session A/B (tools/ab.sh MJH_LIB=...)
set -e
rm -rf build/prev; mkdir -p build/prev/pkg/csrc build/prev/include
for f in $(git ls-tree --name-only ${1:-HEAD} mujoco_warp_amd/csrc/); do git show ${1:-HEAD}:$f > build/prev/pkg/csrc/$(basename $f); done
git show ${1:-HEAD}:include/mjhip.h > build/prev/include/mjhip.h
src=build/prev/pkg/csrc/unity.hip; [ -f $src ] || src=build/prev/pkg/csrc/mjhip.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -Wno-pass-failed -o mujoco_warp_amd/libmjhip_prev.so $src
