#!/bin/bash
# Kernel-trace statistics of the headline bench, temporal-filter / TPL kernels: bash tools/kt_stage.sh -> prints name, calls, average ns
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_stage
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_stage -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline > $REPO/gpurun_out/r3/kt_stage.log 2>&1 || echo "failed"
f=$(find /tmp/kt_stage -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("tf_", "tpl_", "me_b64", "convolve", "wiener", "cdef", "sgr")):
        print(r["Name"][:70].ljust(70), r["Calls"], round(float(r["AverageNs"]) / 1e6, 4), "ms")
PY
