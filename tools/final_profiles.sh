#!/bin/bash
# Round-end measurement on the GPU box (from the repo root):  bash tools/final_profiles.sh <round-tag>   -> gpurun_out/final/*
#   <tag>_bench_4k10.json             the default `python bench.py` line (headline: 4K 10-bit preset 8; live PMC traffic, CPU baselines)
#   <tag>_rocprof_kernel_stats.csv    rocprofv3 --kernel-trace --stats of the same command (without its PMC / CPU legs)
#   <tag>_pmc_instruction_mix.json    SQ counters per kernel (tools/pmc_headline.sh)
#   <tag>_bench_1080p_me.json         configs[1] for comparison with round 1
set -e
REPO=$PWD
TAG=${1:-r02}
OUT=$REPO/gpurun_out/final
mkdir -p $OUT $REPO/gpurun_out/r2
python3 bench.py > $OUT/${TAG}_bench_4k10.json 2> $OUT/bench_4k10.err
echo "headline bench done"
python3 bench.py --workload me1080 --no-cpu-baseline > $OUT/${TAG}_bench_1080p_me.json 2> $OUT/bench_me1080.err || echo "me1080 failed"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_head
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_head -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-pmc --no-cpu-baseline > $OUT/kt_head.log 2>&1 || echo "kernel-trace failed"
f=$(find /tmp/kt_head -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/${TAG}_rocprof_kernel_stats.csv
echo "kernel stats done"
cd $REPO && bash tools/pmc_headline.sh final > /dev/null && cp gpurun_out/r2/pmc_final.json $OUT/${TAG}_pmc_instruction_mix.json
ls -la $OUT
