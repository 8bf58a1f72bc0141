#!/bin/bash
# tools/ab.sh <tag> [<tag> ...]: headline stage times of library builds in ONE GPU call ("main" = libsvtav1_hip.so, any other tag =
# libsvtav1_hip_<tag>.so from tools/mkvariant.sh).  Results under gpurun_out/r3/ab_<tag>.json
mkdir -p gpurun_out/r3
for round in 1 2; do
for tag in "$@"; do
    lib=svt-av1-mod-by-patman_amd/csrc/libsvtav1_hip.so
    [ "$tag" != main ] && lib=svt-av1-mod-by-patman_amd/csrc/libsvtav1_hip_$tag.so
    SVTAV1_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-pmc --no-lf --no-cpu-baseline > gpurun_out/r3/ab_$tag.json 2> gpurun_out/r3/ab_$tag.err || { echo "$tag FAILED"; tail -3 gpurun_out/r3/ab_$tag.err; continue; }
    python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
d = json.load(open(f"gpurun_out/r3/ab_{tag}.json"))
print(f"{tag:12s} ms/step {d['ms_per_step']:.3f}  " + "  ".join(f"{k} {v:.3f}" for k, v in d["stage_ms"].items()))
PY
done
done
