#!/bin/bash
# Vector instructions per wave of the transform kernels per stage: the headline workload with SVTAV1_BENCH_TXMODE = fwd | fwdq | full
# (descriptor flags switch the later stages off).  Usage: bash tools/pmc_txmode.sh  -> gpurun_out/r2/pmc_txmode.json
REPO=$PWD
mkdir -p $REPO/gpurun_out/r2
cd /tmp && export TMPDIR=/tmp
for mode in fwd fwdq full; do
  rm -rf /tmp/pmcm_$mode
  export SVTAV1_BENCH_TXMODE=$mode
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d /tmp/pmcm_$mode -- python3 $REPO/bench.py --pmc-child --steps 2 --warmup 1 --no-cpu-baseline --no-lf > /tmp/pmcm_$mode.log 2>&1 || { echo "pass $mode failed"; tail -5 /tmp/pmcm_$mode.log; }
done
python3 - "$REPO/gpurun_out/r2/pmc_txmode.json" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
out = {}
for mode in ("fwd", "fwdq", "full"):
    acc, disp = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(set))
    for f in glob.glob(f"/tmp/pmcm_{mode}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "txfm_kernel" not in k: continue
            k = k[k.index("txfm_kernel"):k.index(">") + 1]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    out[mode] = {}
    for k, cs in acc.items():
        c = {n: v / max(1, len(disp[k][n])) for n, v in cs.items()}
        w = max(1.0, c.get("SQ_WAVES", 1.0))
        out[mode][k] = {"waves": round(w), **{n + "_per_wave": round(v / w, 1) for n, v in sorted(c.items()) if n != "SQ_WAVES"}}
json.dump(out, open(sys.argv[1], "w"), indent=1); print(json.dumps(out, indent=1))
PY
