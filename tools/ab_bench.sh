#!/bin/bash
# A/B of library builds on the headline workload (no PMC / CPU legs): tools/ab_bench.sh <tag> <lib.so> [<tag> <lib.so> ...]
# Prints the stage times of every build; results under gpurun_out/r2/ab_<tag>.json
set -e
mkdir -p gpurun_out/r2
while [ $# -ge 2 ]; do
    tag=$1; lib=$2; shift 2
    SVTAV1_HIP_LIB=$lib python bench.py --steps 10 --warmup 2 --no-pmc --no-lf --no-cpu-baseline > gpurun_out/r2/ab_$tag.json 2> gpurun_out/r2/ab_$tag.err
    python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
d = json.load(open(f"gpurun_out/r2/ab_{tag}.json"))
print(tag, "value", d["value"], "ms/step", d["ms_per_step"], {r["kernel"]: (r["launch_ms"], r["achieved"]) for r in d["roofline_all"]})
PY
done
