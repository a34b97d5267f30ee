echo "== fhn 8192x1024 (tune_fused, self)"
TUNE_NY=1024 TUNE_STEPS=400 TUNE_ROUNDS=4 TUNE_VARIANTS="remap=0;oneround=1;lockstep=0;oneround=1,lockstep=0;oneround=1,remap=1;oneround=1,strips=2;oneround=1,strips=8;strips=2;chunk=19;chunk=19,lockstep=0" python tools/tune_fused.py 2>&1 | grep median
echo "== fhn 8192x2048"
TUNE_NY=2048 TUNE_STEPS=300 TUNE_ROUNDS=3 TUNE_VARIANTS="remap=0;oneround=1;oneround=1,lockstep=0;remap=2" python tools/tune_fused.py 2>&1 | grep median
