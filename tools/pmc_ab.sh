#!/bin/bash
# Instruction counters of the headline kernels for several library builds: tools/pmc_ab.sh <tag> <lib.so> [<tag> <lib.so> ...]
# -> gpurun_out/r2/pmc_ab_<tag>.json (per kernel, per dispatch)
REPO=$PWD
mkdir -p $REPO/gpurun_out/r2
cd /tmp && export TMPDIR=/tmp
while [ $# -ge 2 ]; do
  tag=$1; export SVTAV1_HIP_LIB=$REPO/$2; shift 2
  rm -rf /tmp/pmcab_$tag
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d /tmp/pmcab_$tag -- python3 $REPO/bench.py --pmc-child --steps 2 --warmup 1 --no-cpu-baseline --no-lf > /tmp/pmcab_$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 /tmp/pmcab_$tag.log; }
  python3 - "$tag" "$REPO/gpurun_out/r2/pmc_ab_$tag.json" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
tag, outp = sys.argv[1], sys.argv[2]
acc, disp = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(set))
for f in glob.glob(f"/tmp/pmcab_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "me_b64" in k: k = "me_b64_kernel"
        elif "txfm_kernel" in k: k = k[k.index("txfm_kernel"):k.index(">") + 1]
        else: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
out = {k: {c: round(v / max(1, len(disp[k][c]))) for c, v in sorted(cs.items())} for k, cs in acc.items()}
json.dump(out, open(outp, "w"), indent=1); print(tag, json.dumps(out.get("me_b64_kernel")))
PY
done
