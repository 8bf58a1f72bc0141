#!/bin/bash
# PMC passes over a bench workload (separate passes; no trace domains besides kernel-trace).
#   tools/run_pmc.sh <tag> <kernel-name-substring> <bench.py args...>   -> gpurun_out/pmc_<tag>_<pass>.json
set -e
REPO=$PWD
TAG=$1; PAT=$2; shift 2
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED" "SQ_WAVES SQ_IFETCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$i -- python3 $REPO/bench.py "$@" > $REPO/gpurun_out/pmc_run_$i.log 2>&1 || echo "pass $i failed"
  python3 $REPO/tools/pmc_summary.py /tmp/pmc_$i "$PAT" > $REPO/gpurun_out/pmc_${TAG}_$i.json || true
done
cat $REPO/gpurun_out/pmc_${TAG}_*.json
