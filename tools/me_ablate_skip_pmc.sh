#!/bin/bash
# VALU / LDS / SALU instructions of me_b64_kernel per (stop stage, skip mask): rocprofv3 --pmc over tools/me_ablate_skip.py.
# Usage (GPU box, repo root, ABLATE=1 build present): bash tools/me_ablate_skip_pmc.sh -> gpurun_out/r3/me_ablate_skip.json
REPO=$PWD
export SVTAV1_HIP_LIB=$REPO/svt-av1-mod-by-patman_amd/csrc/libsvtav1_hip_ablate.so
mkdir -p $REPO/gpurun_out/r3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/abl_skip
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
  --kernel-trace --output-format csv -d /tmp/abl_skip -- python3 $REPO/tools/me_ablate_skip.py > /tmp/abl_skip.log 2>&1 || { tail -20 /tmp/abl_skip.log; exit 1; }
tail -1 /tmp/abl_skip.log
python3 - "$REPO/gpurun_out/r3/me_ablate_skip.json" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
rows = defaultdict(dict)
for f in glob.glob("/tmp/abl_skip/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "me_b64_kernel" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
import os
ST = [int(x) for x in os.environ.get("ME_ABLATE_STOPS", "1,2,3,4,99").split(",")]
MK = [int(x) for x in os.environ.get("ME_ABLATE_MASKS", "0,1,2,3,7").split(",")]
keys = [f"stop{s}_mask{m}" for s in ST for m in MK]
per = len(ids) // len(keys)
out = {}
for k, n in enumerate(keys):
    grp = ids[k * per:(k + 1) * per]
    out[n] = {c: round(sum(rows[i][c] for i in grp) / len(grp)) for c in rows[grp[0]]}
json.dump(out, open(sys.argv[1], "w"), indent=1)
for n in keys:
    print(n, out[n]["SQ_INSTS_VALU"], out[n]["SQ_ACTIVE_INST_VALU"], out[n]["SQ_INSTS_SALU"], out[n]["SQ_INSTS_LDS"])
PY
