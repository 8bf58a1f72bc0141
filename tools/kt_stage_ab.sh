#!/bin/bash
# tools/kt_stage_ab.sh <tag> <lib.so> ...: kernel-trace averages of the TF / TPL kernels for several library builds
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
while [ $# -ge 2 ]; do
  tag=$1; export SVTAV1_HIP_LIB=$REPO/$2; shift 2
  rm -rf /tmp/kt_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline > $REPO/gpurun_out/r2/kt_$tag.log 2>&1 || echo "failed"
  f=$(find /tmp/kt_$tag -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$tag" <<'PY'
import csv, sys
print("==", sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("tf_refine", "tf_predict", "tpl_")):
        print(r["Name"][:70].ljust(70), r["Calls"], round(float(r["AverageNs"]) / 1e6, 4), "ms")
PY
done
