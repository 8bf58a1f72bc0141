#!/bin/bash
# bash tools/ab_me.sh <tagA> <tagB>: me_b64_kernel time of two builds (libsvtav1_hip_<tag>.so; "main" = the in-tree build), alternating, same box
D=$PWD/svt-av1-mod-by-patman_amd/csrc
for i in 1 2; do for t in "$@"; do
  if [ "$t" = main ]; then python3 tools/me_time.py; else SVTAV1_HIP_LIB=$D/libsvtav1_hip_$t.so python3 tools/me_time.py; fi || exit 1
done; done
