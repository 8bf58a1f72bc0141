#!/bin/bash
# Round-end measurement on the GPU box (from the repo root):  bash tools/final_profiles.sh <round-tag>   -> gpurun_out/final/*
#   <tag>_bench_4k10.json             the default `python bench.py` line (headline: 4K 10-bit preset 8; live PMC traffic + issue counters
#                                     for every roofline_all entry, CPU baselines, the encoder-level runs)
#   <tag>_rocprof_kernel_stats.csv    rocprofv3 --kernel-trace --stats of the same command (without its PMC / CPU legs)
#   <tag>_bench_1080p_me.json         configs[1] for comparison with round 1
#   <tag>_bench_gpus2_rehearsal.log   `python bench.py --gpus 2` with no launcher around it: the ranks it starts itself (both on the
#                                     box's one GPU: SVTAV1_BENCH_REHEARSAL=1; the value is not a scaling number)
#   <tag>_enc_hooks_ab.txt            the patched encoder on the x86 intrinsics table with each whole-picture hook switched on alone
#   <tag>_enc_hook_timers.txt         per-hook host time inside the encoder (svt_hip_bind_dev's exit report)
# Each step prints a line when it is done (the box's watchdog wants output every few minutes).
set -e
REPO=$PWD
TAG=${1:-r03}
OUT=$REPO/gpurun_out/final
mkdir -p $OUT
python3 bench.py > $OUT/${TAG}_bench_4k10.json 2> $OUT/bench_4k10.err
echo "headline bench done"
python3 bench.py --workload me1080 --no-cpu-baseline > $OUT/${TAG}_bench_1080p_me.json 2> $OUT/bench_me1080.err || echo "me1080 failed"
echo "1080p bench done"
SVTAV1_BENCH_REHEARSAL=1 python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-lf > $OUT/${TAG}_bench_gpus2_rehearsal.log 2>&1 || echo "gpus2 rehearsal failed"
echo "gpus2 rehearsal done"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_head
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_head -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-pmc --no-cpu-baseline > $OUT/kt_head.log 2>&1 || echo "kernel-trace failed"
f=$(find /tmp/kt_head -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/${TAG}_rocprof_kernel_stats.csv
echo "kernel stats done"
cd $REPO
if [ -x oracle/_ref/e2e/SvtAv1EncApp ]; then
    python3 tools/enc_hooks_ab.py 1080p > $OUT/${TAG}_enc_hooks_ab.txt 2>&1 || echo "enc_hooks_ab failed"
    echo "encoder hook A/B done"
    bash tools/hooktime.sh > $OUT/${TAG}_enc_hook_timers.txt 2>&1 || echo "hooktime failed"
    echo "encoder hook timers done"
fi
ls -la $OUT
