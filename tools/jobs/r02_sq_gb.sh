OUT=$GRAFT_REPO_ROOT/gpurun_out/r02/sq; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CRD_AUTOTUNE=0 MODEL=goldbeter NX=4096 NY=4096 STEPS=60
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT/sq_gb4096 -- python3 $GRAFT_REPO_ROOT/tools/slab_run.py > $OUT/sq_gb4096.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/sq2_gb4096 -- python3 $GRAFT_REPO_ROOT/tools/slab_run.py > $OUT/sq2_gb4096.log 2>&1
MODEL=fhn rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT/sq_fhn4096 -- python3 $GRAFT_REPO_ROOT/tools/slab_run.py > $OUT/sq_fhn4096.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('/root/repo/gpurun_out/r02/sq/sq*4096/')):
    agg=collections.defaultdict(list)
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'fused' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(d.split('/')[-2], {k: round(sum(v)/len(v)) for k,v in agg.items()}, 'launches', max((len(v) for v in agg.values()), default=0))
PY
