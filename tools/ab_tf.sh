#!/bin/bash
# tools/ab_tf.sh <tag> <lib.so> ...: temporal-filter stage time and result checksum of the headline bench for several library builds
mkdir -p gpurun_out/r3
while [ $# -ge 2 ]; do
    tag=$1; lib=$2; shift 2
    SVTAV1_HIP_LIB=$lib python bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline > gpurun_out/r3/abtf_$tag.json 2> gpurun_out/r3/abtf_$tag.err
    python - "$tag" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r3/abtf_{sys.argv[1]}.json"))
for r in d["roofline_all"]:
    if "tf_" in r["kernel"] or "tpl" in r["kernel"]:
        print(sys.argv[1], r["kernel"][:40], r["launch_ms"], r.get("result_checksum"))
PY
done
