#!/bin/bash
# Round-end measurement: bench lines, rocprofv3 kernel-trace stats and PMC traffic for the three workloads.
# Usage (on the GPU box, from the repo root): bash tools/final_profiles.sh <round-tag>   -> gpurun_out/final/*
set -e
REPO=$PWD
TAG=${1:-r01}
OUT=$REPO/gpurun_out/final
mkdir -p $OUT
python3 bench.py > $OUT/${TAG}_bench_1080p_me.json 2> $OUT/bench_me.err
python3 bench.py --workload txfm > $OUT/${TAG}_bench_4k10_txfm.json 2> $OUT/bench_txfm.err
python3 bench.py --workload lf > $OUT/${TAG}_bench_4k10_lf.json 2> $OUT/bench_lf.err
cd /tmp && export TMPDIR=/tmp
for wl in me txfm lf; do
  rm -rf /tmp/kt_$wl
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$wl -- python3 $REPO/bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline > $OUT/kt_$wl.log 2>&1 || echo "kernel-trace $wl failed"
  f=$(find /tmp/kt_$wl -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_rocprof_kernel_stats_$wl.csv
done
# PMC: HBM-side traffic of the dominant kernels (separate passes, nothing but --kernel-trace next to --pmc)
for wl in me txfm; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${wl}_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${wl}_$c -- python3 $REPO/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_${wl}_$c.log 2>&1 || echo "pmc $wl $c failed"
    python3 $REPO/tools/pmc_summary.py /tmp/pmc_${wl}_$c $( [ $wl = me ] && echo me_b64 || echo txfm_kernel ) > $OUT/${TAG}_pmc_${wl}_$c.json || true
  done
done
ls -la $OUT
