# Round 3 (late): does the tuner (bursts + sustained final) pick what sustained stepping ranks first?  bench (tuner) twice per
# configuration beside tools/tune_fused.py (real stepping, variants interleaved in one process).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03/exp_tuner_fidelity; mkdir -p $OUT
cd $R
V=""; for o in 0 1; do for m in 0 1 2; do for c in 1 2; do for n in 0 1; do [ "$o$m" = "12" ] && continue; V="$V;oneround=$o,remap=$m,cols=$c,nt=$n"; done; done; done; done
tf() { local tag=$1; shift; echo "== $tag (tune_fused, sustained)" | tee -a $OUT/fidelity.txt; env TUNE_VARIANTS="${V#;}" TUNE_ROUNDS=4 "$@" timeout -k 10 400 python3 tools/tune_fused.py 2>&1 | grep median | sort -k3 -n | head -8 | tee -a $OUT/fidelity.txt; }
b() { local name=$1; shift; for rep in 1 2; do CRD_AUTOTUNE_VERBOSE=1 timeout -k 10 400 python3 bench.py --no-cpu-baseline --staged-steps 0 "$@" > $OUT/$name.$rep.json 2> $OUT/$name.$rep.err; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT $name; exit 1; fi
  python3 - $OUT/$name.$rep.json <<'PY' | tee -a $OUT/fidelity.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']; p=d['config']['launch_plan']
print(sys.argv[1].split('/')[-1], 'bench: ms/step %.4f frac %.3f plan oneround=%d,remap=%d,cols=%d,nt=%d tuner: default %.4f chosen %.4f' % (d['ms_per_step'], r['frac'], p['one_round'], p['xcd_mapping'], p['columns_per_lane'], p['nontemporal_stores'], p['ms_default'], p['ms_chosen']))
PY
  grep "final" $OUT/$name.$rep.err | sed 's/libcrd autotune: //' >> $OUT/fidelity.txt
done; }
tf "goldbeter 4096" TUNE_SIZE=4096 TUNE_MODEL=goldbeter TUNE_STEPS=400
b C4_gb4096 --size 4096 --model goldbeter
tf "fhn 4096" TUNE_SIZE=4096 TUNE_STEPS=400
b C2_4096 --size 4096
tf "fhn 8192" TUNE_SIZE=8192 TUNE_STEPS=150
b C3_8192
tf "fhn f32 8192" TUNE_SIZE=8192 TUNE_PRECISION=f32 TUNE_STEPS=200
b F32_8192 --precision f32
tf "fhn 8192x1024" TUNE_SIZE=8192 TUNE_NY=1024 TUNE_STEPS=800
CRD_AUTOTUNE_VERBOSE=1 NYS=1024 VARIANTS=self,rccl:0 timeout -k 10 300 python3 tools/ring_overhead.py > $OUT/ring_overhead_verbose.txt 2>&1
grep "ny=\|final" $OUT/ring_overhead_verbose.txt | sed 's/libcrd autotune: //' | tee -a $OUT/fidelity.txt
