# Round 5: issue-side counters of the step kernel's one- and two-step instantiations (what bounds the two-step launch, now that it
# has left the HBM roof): SQ_* per launch, for the plans in $2, on the configuration $1 ("model precision size").
# Lands in gpurun_out/r05/sq/<config>.json
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05/sq; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
set -- $1 "$2"; model=$1; prec=$2; size=$3; plans=$4; name=${model}_${prec}_${size}
W=/tmp/sq_$name; rm -rf $W; mkdir -p $W
for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 400 rocprofv3 --pmc $pass --output-format csv -d $W/$tag -- python3 $R/tools/plan_sweep.py --model $model --precision $prec --size $size --plans "$plans" --out $W/plans_$tag.json > $W/$tag.log 2>&1 || { tail -5 $W/$tag.log; exit 1; }
  python3 $R/tools/plan_sweep_summary.py --plans $W/plans_$tag.json --counters $(find $W/$tag -name "*counter_collection.csv" | head -1) > $OUT/${name}_$tag.json || exit 1
done
