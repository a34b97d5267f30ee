# Round-3 baseline on today's box: autotune tables for every configuration, ring rehearsal, tuner variants on the 8-GPU share.
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03/baseline; mkdir -p $OUT
cd $R
export CRD_AUTOTUNE_VERBOSE=1
python bench.py --no-cpu-baseline --staged-steps 0 > $OUT/C3.json 2> $OUT/C3.err
python bench.py --size 4096 --no-cpu-baseline --staged-steps 0 > $OUT/C2.json 2> $OUT/C2.err
python bench.py --size 4096 --model goldbeter --no-cpu-baseline --staged-steps 0 > $OUT/C4.json 2> $OUT/C4.err
python bench.py --model goldbeter --no-cpu-baseline --staged-steps 0 --steps 100 > $OUT/GB8192.json 2> $OUT/GB8192.err
python bench.py --precision f32 --no-cpu-baseline --staged-steps 0 > $OUT/F32_8192.json 2> $OUT/F32_8192.err
python bench.py --size 16384 --precision f32 --no-cpu-baseline --steps 60 --staged-steps 0 > $OUT/C5.json 2> $OUT/C5.err
NYS=1024 python tools/ring_overhead.py > $OUT/ring_overhead.txt 2>&1
unset CRD_AUTOTUNE_VERBOSE
grep -h "autotune" $OUT/*.err $OUT/ring_overhead.txt | sort | uniq -c | sort -k1,1nr | head -0
for f in $OUT/*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[1].split('/')[-1], 'ms/step %.4f value %.3e kernel_ms %.4f frac %.3f plan %s' % (d['ms_per_step'], d['value'], r['kernel_ms'], r['frac'], {k:d['config']['launch_plan'][k] for k in ('tuned','one_round','xcd_mapping')}))
PY
done
grep "ny=" $OUT/ring_overhead.txt
TUNE_NY=1024 TUNE_STEPS=400 TUNE_VARIANTS="chunk=0;oneround=1;oneround=1,remap=1;remap=1;remap=2;chunk=36;chunk=40;chunk=44" python tools/tune_fused.py > $OUT/tune_8192x1024.txt 2>&1; cat $OUT/tune_8192x1024.txt
TUNE_SIZE=4096 TUNE_STEPS=400 TUNE_VARIANTS="chunk=0;oneround=1;oneround=1,remap=1;remap=1;remap=2;chunk=64;chunk=64,remap=1;chunk=78;chunk=78,remap=1" python tools/tune_fused.py > $OUT/tune_4096.txt 2>&1; cat $OUT/tune_4096.txt
TUNE_SIZE=4096 TUNE_MODEL=goldbeter TUNE_STEPS=400 TUNE_VARIANTS="chunk=0;oneround=1;oneround=1,remap=1;remap=1;remap=2;chunk=64;chunk=64,remap=1;chunk=64,remap=2;chunk=78,remap=1;chunk=128,remap=1" python tools/tune_fused.py > $OUT/tune_gb4096.txt 2>&1; cat $OUT/tune_gb4096.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
