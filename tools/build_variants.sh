#!/bin/bash
# Builds libsvtav1_hip.so and the tuning variants (ABLATE=1 -> libsvtav1_hip_ablate.so, PROF=1 -> libsvtav1_hip_prof.so with "prof" arg).
set -e
D=$(cd "$(dirname "$0")/../svt-av1-mod-by-patman_amd/csrc" && pwd)
cd "$D"
rm -f me_frame.o me_kernels.o
make -j8 ABLATE=1 2>&1 | grep -i "error" && exit 1
cp libsvtav1_hip.so libsvtav1_hip_ablate.so
if [ "$1" = prof ]; then
  rm -f me_frame.o me_kernels.o
  make -j8 PROF=1 2>&1 | grep -i "error" && exit 1
  cp libsvtav1_hip.so libsvtav1_hip_prof.so
fi
rm -f me_frame.o me_kernels.o
make -j8 2>&1 | grep -i "error" && exit 1
ls -la --time-style=full-iso "$D"/*.so | cut -c30-
