#!/bin/bash
# Instruction counters of the temporal-filter / TPL kernels inside the headline bench (its roofline_all legs):
#   [PMC_STAGE_REGEX='cdef_|sgr_|dlf_|wiener_'] bash tools/pmc_stage.sh -> gpurun_out/r2/pmc_stage.json
REPO=$PWD
mkdir -p $REPO/gpurun_out/r2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcst
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-include-regex "${PMC_STAGE_REGEX:-tf_|tpl_}" --kernel-trace --output-format csv -d /tmp/pmcst -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-pmc --no-cpu-baseline > $REPO/gpurun_out/r2/pmcst_run.log 2>&1 || { echo "pass failed"; tail -5 /tmp/pmcst.log; }
python3 - "$REPO/gpurun_out/r2/pmc_stage.json" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
acc, disp, dur = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(set)), defaultdict(list)
def key(k):
    import re
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", k)
    return m.group(1) if m else None
for f in glob.glob("/tmp/pmcst/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = key(r["Kernel_Name"])
        if k: acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for f in glob.glob("/tmp/pmcst/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = key(r["Kernel_Name"])
        if k: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
out = {k: {**{c: round(v / max(1, len(disp[k][c]))) for c, v in sorted(cs.items())}, "ms_under_pmc": round(sum(dur[k]) / max(1, len(dur[k])), 4), "dispatches": len(dur[k])} for k, cs in acc.items()}
json.dump(out, open(sys.argv[1], "w"), indent=1); print(json.dumps(out, indent=1))
PY
