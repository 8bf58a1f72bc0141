#!/bin/bash
# Instruction counts of me_b64_kernel truncated behind each stage (ABLATE=1 build): one rocprofv3 --pmc pass over
# tools/me_ablate.py; prints per-stage averages.   Usage (GPU box, repo root): bash tools/me_ablate_pmc.sh [--1080]
REPO=$PWD
export SVTAV1_HIP_LIB=$REPO/svt-av1-mod-by-patman_amd/csrc/libsvtav1_hip_ablate.so
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/abl_pmc
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU \
  --kernel-trace --output-format csv -d /tmp/abl_pmc -- python3 $REPO/tools/me_ablate.py "$@" > /tmp/abl_pmc.log 2>&1 || { tail -20 /tmp/abl_pmc.log; exit 1; }
python3 - <<'PY'
import csv, glob, json
from collections import defaultdict
rows = defaultdict(dict)
for f in glob.glob("/tmp/abl_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "me_b64_kernel" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
names = ["init+sources", "zz_sad", "pre-HME", "HME L0", "HME L1", "HME L2", "centre/prune", "full-pel", "all"]
per = len(ids) // 9
out, prev = {}, defaultdict(float)
for k, n in enumerate(names):
    grp = ids[k * per:(k + 1) * per]
    avg = {c: sum(rows[i][c] for i in grp) / len(grp) for c in rows[grp[0]]}
    out[n] = {c: round(avg[c] - prev[c]) for c in avg}
    prev = avg
out["total"] = {c: round(v) for c, v in prev.items()}
print(json.dumps({"me_ablation_pmc_per_launch": out}))
PY
