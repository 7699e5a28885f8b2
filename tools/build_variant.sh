#!/bin/bash
# Build the working tree as mujoco_warp_amd/libmjhip_<tag>.so with extra compiler flags, for same-session A/B runs
# (MJH_LIB=mujoco_warp_amd/libmjhip_<tag>.so):  tools/build_variant.sh <tag> [-DMACRO=value ...]
set -e
TAG=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -fvisibility=hidden -fno-slp-vectorize -Wno-unused-value -Wno-pass-failed "$@" \
  -o mujoco_warp_amd/libmjhip_$TAG.so mujoco_warp_amd/csrc/unity.hip
