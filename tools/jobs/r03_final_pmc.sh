# Round-3 record, second call: PMC traffic of the launch plans in use (separate --pmc passes, as the guide prescribes) and the
# issue-side counters of the Goldbeter instantiation.  Plans without non-temporal stores are pinned with the tuning knobs, plans with
# them through bench.py --launch-plan.  Lands in gpurun_out/r03/final/pmc/.
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03/final; mkdir -p $OUT/pmc
# PMC traffic of the plans in use (separate --pmc passes, as the guide prescribes; plan pinned with the tuning knobs)
cd /tmp
export CRD_TUNING=1
pmc_pair() { # name, points, bench args...
  local name=$1 pts=$2; shift 2
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc/${name}_$ctr -- python3 $R/bench.py --steps 30 --warmup 5 --preheat-ms 0 --no-cpu-baseline --staged-steps 0 "$@" > $OUT/pmc/${name}_$ctr.log 2>&1
  done
  python3 $R/tools/pmc_summary.py $(find $OUT/pmc/${name}_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $OUT/pmc/${name}_WRITE_SIZE -name "*counter_collection.csv" | head -1) --points $pts --match ${MATCH:-fused} > $OUT/pmc/traffic_$name.json
  cat $OUT/pmc/traffic_$name.json; rm -rf $OUT/pmc/${name}_FETCH_SIZE $OUT/pmc/${name}_WRITE_SIZE
}
if [ "${PMC_SET:-all}" = "all" ]; then   # PMC_SET=nt: only the plans with non-temporal stores (the others were recorded by an earlier call)
CRD_FUSED_REMAP=0 CRD_FUSED_COLS=1 pmc_pair fhn_f64_map0 67108864
CRD_FUSED_REMAP=2 CRD_FUSED_COLS=1 pmc_pair fhn_f64_map2 67108864
CRD_FUSED_REMAP=1 CRD_FUSED_COLS=2 pmc_pair fhn_f32_16384_map1_cols2 268435456 --size 16384 --precision f32
CRD_FUSED_REMAP=0 CRD_FUSED_COLS=1 pmc_pair fhn_f32_16384_map0_cols1 268435456 --size 16384 --precision f32
CRD_FUSED_REMAP=1 CRD_FUSED_COLS=2 CRD_FUSED_ONEROUND=1 pmc_pair goldbeter_f64_4096_oneround_map1_cols2 16777216 --size 4096 --model goldbeter
CRD_FUSED_REMAP=1 CRD_FUSED_COLS=1 pmc_pair fhn_f64_map1 67108864
CRD_FUSED_REMAP=1 CRD_FUSED_COLS=1 pmc_pair fhn_f32_8192_map1_cols1 67108864 --precision f32
CRD_FUSED_REMAP=2 CRD_FUSED_COLS=1 pmc_pair fhn_f32_8192_map2_cols1 67108864 --precision f32
CRD_FUSED_REMAP=0 CRD_FUSED_COLS=2 pmc_pair fhn_f32_16384_map0_cols2 268435456 --size 16384 --precision f32
CRD_FUSED_REMAP=2 CRD_FUSED_COLS=2 pmc_pair fhn_f32_16384_map2_cols2 268435456 --size 16384 --precision f32
CRD_FUSED_REMAP=0 CRD_FUSED_COLS=1 pmc_pair goldbeter_f64_map0 67108864 --model goldbeter
CRD_FUSED_REMAP=2 CRD_FUSED_COLS=1 pmc_pair goldbeter_f64_map2 67108864 --model goldbeter
# issue-side counters of the Goldbeter instantiation at 4096^2: one column per lane on the plain plan against two columns on one-round chunks
for v in "cols1 0 1" "cols2 1 2"; do set -- $v
  CRD_FUSED_ONEROUND=$2 CRD_FUSED_COLS=$3 CRD_FUSED_REMAP=0 MODEL=goldbeter NX=4096 NY=4096 STEPS=60 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc/sq_gb4096_$1 -- python3 $R/tools/slab_run.py > $OUT/pmc/sq_gb4096_$1.log 2>&1
done
python3 - <<'PY' > $OUT/pmc/sq_goldbeter_4096.json
import csv, glob, collections, json, os
out = {}
for d in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03/final/pmc/sq_gb4096_*/")):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fused" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    rec = {k: sum(v) / len(v) for k, v in agg.items()}
    rec["launches_sampled"] = max((len(v) for v in agg.values()), default=0)
    if "SQ_INSTS_VALU" in rec:
        rec["valu_wave_instructions_per_grid_point"] = rec["SQ_INSTS_VALU"] / (4096.0 * 4096.0)
    out[d.rstrip("/").split("_")[-1]] = rec
print(json.dumps(out, indent=1))
PY
cat $OUT/pmc/sq_goldbeter_4096.json; rm -rf $OUT/pmc/sq_gb4096_cols1 $OUT/pmc/sq_gb4096_cols2
fi
unset CRD_TUNING CRD_FUSED_REMAP CRD_FUSED_COLS CRD_FUSED_ONEROUND
if [ "${PMC_SET:-all}" != "staged" ]; then
# plans with non-temporal stores of the new state (late round 3), pinned through the API
pmc_pair fhn_f64_map0_nt 67108864 --launch-plan 0,0,1,1
pmc_pair fhn_f64_map1_nt 67108864 --launch-plan 0,1,1,1
pmc_pair fhn_f64_map2_nt 67108864 --launch-plan 0,2,1,1
pmc_pair fhn_f64_map1_cols2_nt 67108864 --launch-plan 0,1,2,1
pmc_pair goldbeter_f64_map0_nt 67108864 --model goldbeter --launch-plan 0,0,1,1
pmc_pair goldbeter_f64_map0_cols2_nt 67108864 --model goldbeter --launch-plan 2,0,2,1
pmc_pair fhn_f32_16384_map0_cols2_nt 268435456 --size 16384 --precision f32 --launch-plan 0,0,2,1
pmc_pair fhn_f32_16384_map1_cols2_nt 268435456 --size 16384 --precision f32 --launch-plan 0,1,2,1
pmc_pair fhn_f32_16384_map0_cols1_nt 268435456 --size 16384 --precision f32 --launch-plan 0,0,1,1
pmc_pair fhn_f32_16384_map1_cols1_nt 268435456 --size 16384 --precision f32 --launch-plan 0,1,1,1
fi
# the staged stepper's kernels (non-temporal stores of their results on slabs of 32 MiB per plane and more)
MATCH=stage_kernel pmc_pair staged_fhn_f64_8192 67108864 --stepper staged
