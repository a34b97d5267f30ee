set -x
mkdir -p gpurun_out/r02
tools/rcp_probe > gpurun_out/r02/rcp_probe.txt 2>&1; cat gpurun_out/r02/rcp_probe.txt
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "goldbeter or golden or rk4_trajectory or fp32 or C4" > gpurun_out/r02/pytest_gpu_5.log 2>&1; tail -5 gpurun_out/r02/pytest_gpu_5.log
L="new=crdmodel_amd/libcrd.so;nr1=tools/_variants/libcrd_nr1.so;old2q=tools/_variants/libcrd_gb2q.so"
AB_LIBS="$L" AB_MODEL=goldbeter AB_SIZE=4096 python tools/ab_libs.py
CRD_TUNING=1 CRD_FUSED_CHUNK=64 AB_LIBS="$L" AB_MODEL=goldbeter AB_SIZE=4096 python tools/ab_libs.py
AB_LIBS="$L" AB_MODEL=goldbeter AB_SIZE=8192 AB_STEPS=100 python tools/ab_libs.py
CRD_TUNING=1 CRD_FUSED_CHUNK=64 AB_LIBS="$L" AB_MODEL=goldbeter AB_SIZE=8192 AB_STEPS=100 python tools/ab_libs.py
