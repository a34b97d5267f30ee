# Round 3 (late): cold clocks?  The same timed region behind a short and a long warm-up.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03/exp_warm_clocks; mkdir -p $OUT
cd $R
b() { local name=$1; shift; timeout -k 10 400 python3 bench.py --no-cpu-baseline --staged-steps 0 "$@" > $OUT/$name.json 2> $OUT/$name.err; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT $name; exit 1; fi
  python3 - $OUT/$name.json "$*" <<'PY' | tee -a $OUT/warm.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']; p=d['config']['launch_plan']
print('%-60s ms/step %.4f kernel_ms %.4f frac %.3f plan %d,%d,%d,%d' % (sys.argv[2], d['ms_per_step'], r['kernel_ms'], r['frac'], p['one_round'], p['xcd_mapping'], p['columns_per_lane'], p['nontemporal_stores']))
PY
}
b a --size 4096 --model goldbeter --launch-plan 1,0,1,1
b b --size 4096 --model goldbeter --launch-plan 1,0,1,1 --warmup 3000
b c --size 4096 --model goldbeter --launch-plan 1,0,1,1 --warmup 3000 --steps 1000
b d --size 4096 --launch-plan 0,1,1,1
b e --size 4096 --launch-plan 0,1,1,1 --warmup 3000
b f --launch-plan 0,0,1,1 --steps 20 --warmup 5
b g --launch-plan 0,0,1,1 --steps 20 --warmup 800
b h --launch-plan 0,0,1,1
b i --launch-plan 0,0,1,1 --warmup 800
