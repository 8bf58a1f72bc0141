#!/bin/bash
# PMC passes over the headline workload's kernels (child mode of bench.py).  Usage: bash tools/pmc_headline.sh <tag>
# -> gpurun_out/r2/pmc_<tag>.json : per kernel, per-dispatch averages of each counter.
REPO=$PWD
TAG=${1:-x}
mkdir -p $REPO/gpurun_out/r2
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" \
           "SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT SQ_WAVES_EQ_64 SQ_LEVEL_WAVES"; do
  i=$((i+1)); rm -rf /tmp/pmch_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmch_$i -- python3 $REPO/bench.py --pmc-child --steps 2 --warmup 1 --no-cpu-baseline --no-lf > /tmp/pmch_$i.log 2>&1 || { echo "pass $i failed"; tail -5 /tmp/pmch_$i.log; }
done
python3 - "$REPO/gpurun_out/r2/pmc_$TAG.json" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
acc, disp = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(set))
for f in glob.glob("/tmp/pmch_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "me_b64" in k: k = "me_b64_kernel"
        elif "txfm_kernel" in k: k = k[k.index("txfm_kernel"):k.index(">") + 1]
        else: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
out = {k: {c: round(v / max(1, len(disp[k][c]))) for c, v in sorted(cs.items())} for k, cs in acc.items()}
json.dump(out, open(sys.argv[1], "w"), indent=1); print(json.dumps(out))
PY
