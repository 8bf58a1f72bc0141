set -e
mkdir -p gpurun_out/r3/tfpmc
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  echo "pass $grp"
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/r3/tfpmc/$tag -- python3 $R/bench.py --pmc-child --no-cpu-baseline > $R/gpurun_out/r3/tfpmc/$tag.log 2>&1 || echo "rc $?"
done
python3 - <<'P'
import csv,glob,collections,os
R=os.environ['GRAFT_REPO_ROOT']
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob(R+'/gpurun_out/r3/tfpmc/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:60]
        if 'tf_' in k or 'me_b64' in k:
            acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k in acc:
    print(k)
    for c,v in sorted(acc[k].items()): print('   ',c,v)
P
