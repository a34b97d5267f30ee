for i in 1 2 3; do CRD_AUTOTUNE_VERBOSE=1 python bench.py --no-cpu-baseline --staged-steps 0 --steps 100 2>&1 >/dev/null | grep autotune; echo; done
CRD_AUTOTUNE_VERBOSE=1 python bench.py --no-cpu-baseline --staged-steps 0 --size 4096 2>&1 >/dev/null | grep autotune
CRD_AUTOTUNE_VERBOSE=1 python bench.py --no-cpu-baseline --staged-steps 0 --size 4096 --model goldbeter 2>&1 >/dev/null | grep autotune
echo "== interleaved tune_fused on this box"
TUNE_ROUNDS=4 TUNE_VARIANTS="remap=0;remap=1;remap=2" python tools/tune_fused.py 2>&1 | grep median
