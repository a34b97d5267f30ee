# Round 6: every launch plan the tuner can choose, profiled -- HBM traffic (separate --pmc FETCH_SIZE / WRITE_SIZE passes) and launch
# durations (--kernel-trace) from ONE process that pins the plans in turn (tools/plan_sweep.py), for the configurations in CONFIGS.
# Lands in gpurun_out/r06/sweep/<config>.json (+ the un-profiled pass's event timings).   usage: bash tools/jobs/r06_sweep.sh "fhn f64 8192" ...
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06/sweep; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for cfg in "$@"; do
  set -- $cfg; model=$1; prec=$2; size=$3; name=${model}_${prec}_${size}
  W=/tmp/sweep_$name; rm -rf $W; mkdir -p $W
  timeout -k 10 300 python3 $R/tools/plan_sweep.py --model $model --precision $prec --size $size --out $W/plans_plain.json > $OUT/$name.unprofiled.log 2>&1 || exit 1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --pmc $ctr --output-format csv -d $W/$ctr -- python3 $R/tools/plan_sweep.py --model $model --precision $prec --size $size --out $W/plans_$ctr.json > $W/$ctr.log 2>&1 || { tail -5 $W/$ctr.log; exit 1; }
  done
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $W/trace -- python3 $R/tools/plan_sweep.py --model $model --precision $prec --size $size --out $W/plans_trace.json > $W/trace.log 2>&1 || { tail -5 $W/trace.log; exit 1; }
  python3 $R/tools/plan_sweep_summary.py --plans $W/plans_trace.json --fetch $(find $W/FETCH_SIZE -name "*counter_collection.csv" | head -1) \
      --write $(find $W/WRITE_SIZE -name "*counter_collection.csv" | head -1) --trace $(find $W/trace -name "*kernel_trace.csv" | head -1) > $OUT/$name.json || exit 1
  cp $W/plans_plain.json $OUT/$name.unprofiled.json
  echo "sweep $name done"; date
done
