# tools/ldspmc.sh — LDS cycles per LDS instruction of every kernel of the headline bench (a value near 64 = accesses that are not
# naturally aligned, tools/ubench/lds_unaligned.hip).  GPU box, from the repo root; prints a table.
set -e
mkdir -p gpurun_out/r3/ldspmc
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/r3/ldspmc/a -- python3 $R/bench.py --pmc-child --no-cpu-baseline > $R/gpurun_out/r3/ldspmc/a.log 2>&1 || echo "rc $?"
python3 - <<'P'
import csv,glob,collections,os
R=os.environ['GRAFT_REPO_ROOT']
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(R+'/gpurun_out/r3/ldspmc/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:70]][r['Counter_Name']]+=float(r['Counter_Value'])
print("%-72s %12s %12s %8s %8s" % ("kernel","LDS insts","LDS cycles","cyc/inst","conflict"))
for k,v in sorted(acc.items(), key=lambda kv:-kv[1].get('SQ_LDS_IDX_ACTIVE',0)):
    n=v.get('SQ_INSTS_LDS',0)
    if n: print("%-72s %12.0f %12.0f %8.1f %8.2f" % (k,n,v['SQ_LDS_IDX_ACTIVE'],v['SQ_LDS_IDX_ACTIVE']/n,v['SQ_LDS_BANK_CONFLICT']/n))
P
