#!/bin/bash
# Instruction-cache counters of the headline kernels, mixed and single transform type: bash tools/pmc_icache.sh -> gpurun_out/r2/pmc_icache.json
REPO=$PWD
mkdir -p $REPO/gpurun_out/r2
cd /tmp && export TMPDIR=/tmp
for mode in mix dct; do
  rm -rf /tmp/pmcic_$mode
  export SVTAV1_BENCH_TXTYPE=$mode
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmcic_$mode -- python3 $REPO/bench.py --pmc-child --steps 2 --warmup 1 --no-cpu-baseline --no-lf > /tmp/pmcic_$mode.log 2>&1 || { echo "pass $mode failed"; tail -5 /tmp/pmcic_$mode.log; }
done
python3 - "$REPO/gpurun_out/r2/pmc_icache.json" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
out = {}
for mode in ("mix", "dct"):
    acc, disp, dur = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(set)), defaultdict(list)
    for f in glob.glob(f"/tmp/pmcic_{mode}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "me_b64" in k: k = "me_b64_kernel"
            elif "txfm_kernel" in k: k = k[k.index("txfm_kernel"):k.index(">") + 1]
            else: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    for f in glob.glob(f"/tmp/pmcic_{mode}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "me_b64" in k: k = "me_b64_kernel"
            elif "txfm_kernel" in k: k = k[k.index("txfm_kernel"):k.index(">") + 1]
            else: continue
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    out[mode] = {k: {**{c: round(v / max(1, len(disp[k][c]))) for c, v in sorted(cs.items())}, "ms_under_pmc": round(sum(dur[k]) / max(1, len(dur[k])), 4)} for k, cs in acc.items()}
json.dump(out, open(sys.argv[1], "w"), indent=1); print(json.dumps(out, indent=1))
PY
