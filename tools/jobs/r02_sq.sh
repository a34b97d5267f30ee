OUT=$GRAFT_REPO_ROOT/gpurun_out/r02/sq; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CRD_AUTOTUNE=0
for ny in 1024 8192; do
NY=$ny STEPS=60 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT/sq_$ny -- python3 $GRAFT_REPO_ROOT/tools/slab_run.py > $OUT/sq_$ny.log 2>&1
NY=$ny STEPS=60 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2_$ny -- python3 $GRAFT_REPO_ROOT/tools/slab_run.py > $OUT/sq2_$ny.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('/root/repo/gpurun_out/r02/sq/sq*_*/')):
    agg=collections.defaultdict(list)
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'fused' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(d.split('/')[-2], {k: round(sum(v)/len(v)) for k,v in agg.items()}, 'launches', max((len(v) for v in agg.values()), default=0))
PY
