#!/bin/bash
# tools/ab_env.sh <tag> "<ENV=VAL ...>" ... : the headline workload under different environments (no PMC / CPU legs)
mkdir -p gpurun_out/r2
while [ $# -ge 2 ]; do
    tag=$1; envs=$2; shift 2
    env $envs python bench.py --steps 10 --warmup 2 --no-pmc --no-lf --no-cpu-baseline > gpurun_out/r2/ab_$tag.json 2> gpurun_out/r2/ab_$tag.err
    python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
d = json.load(open(f"gpurun_out/r2/ab_{tag}.json"))
print(tag, "ms/step", d["ms_per_step"], {r["kernel"].replace("txfm_kernel", "tx"): r["launch_ms"] for r in d["roofline_all"]})
PY
done
