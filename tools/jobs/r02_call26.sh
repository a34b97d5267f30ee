timeout -k 10 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
NYS=1024 STEPS=20 python tools/ring_overhead.py 2>&1 | grep "ny="
NYS=1024 STEPS=400 python tools/ring_overhead.py 2>&1 | grep "ny="
