# Round 6: issue-side counters of the step kernel with LDS and barrier waits (the three-step block-strip kernel beside the two-step one):
# SQ_* per launch for the plans in $2 on the configuration $1 ("model precision size"); $3 = tag of the library (CRD_LIBRARY is honoured).
# Lands in gpurun_out/r06/sq/<config>_<tag>_<first counter>.json.  A pass whose counters this device does not have is skipped.
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06/sq; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
tagname=${3:-tree}
set -- $1 "$2"; model=$1; prec=$2; size=$3; plans=$4; name=${model}_${prec}_${size}_${tagname}
W=/tmp/sq_$name; rm -rf $W; mkdir -p $W
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $OUT/sq_counters_available.txt || true
for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
            "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 400 rocprofv3 --pmc $pass --output-format csv -d $W/$tag -- python3 $R/tools/plan_sweep.py --model $model --precision $prec --size $size --warm 6 --steps 24 --plans "$plans" --out $W/plans_$tag.json > $W/$tag.log 2>&1 || { tail -5 $W/$tag.log; continue; }
  python3 $R/tools/plan_sweep_summary.py --plans $W/plans_$tag.json --counters $(find $W/$tag -name "*counter_collection.csv" | head -1) > $OUT/${name}_$tag.json || true
done
