#!/bin/bash
# tools/mkvariant.sh <tag> <dir with replacement sources (or "-")> [extra hipcc flags]
# Builds svt-av1-mod-by-patman_amd/csrc/libsvtav1_hip_<tag>.so: me_frame.hip / me_kernels.hip recompiled with the extra flags (and, if a
# directory is given, with the headers / sources found there taking precedence), every other object as in the main build.
# A tuning aid: SVTAV1_HIP_LIB=<that .so> makes tests and bench.py load it (several builds compared in ONE GPU call).
set -e
tag=$1; dir=$2; shift 2
D=$(cd "$(dirname "$0")/../svt-av1-mod-by-patman_amd/csrc" && pwd)
cd "$D"
make -j8 >/dev/null
inc=""; src_me=me_frame.hip; src_mk=me_kernels.hip
if [ "$dir" != "-" ]; then
  inc="-I$dir"; [ -f "$dir/me_frame.hip" ] && src_me="$dir/me_frame.hip"; [ -f "$dir/me_kernels.hip" ] && src_mk="$dir/me_kernels.hip"
fi
FLAGS="-O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wno-unused-function --offload-arch=gfx950 $inc -I$D -I$D/../../include"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $src_me -o /tmp/me_frame_$tag.o &
/opt/rocm/bin/hipcc $FLAGS "$@" -c $src_mk -o /tmp/me_kernels_$tag.o &
wait
objs=$(ls *.o | grep -v -e '^me_frame.o$' -e '^me_kernels.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libsvtav1_hip_$tag.so $objs /tmp/me_frame_$tag.o /tmp/me_kernels_$tag.o
ls -la libsvtav1_hip_$tag.so
