#!/bin/bash
# The reference's own work flow on its shipped FHN parameter set, end to end on one GPU: solver -> subdomain files -> frames and
# torus mapping.  Writes under gpurun_out/demo (scratch); a few frames are kept under docs/.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
D=$R/gpurun_out/demo
rm -rf "$D" && mkdir -p "$D"
sed 's/^outputTimestep.*/outputTimestep = 10/' "$R/tests/golden/ini/fhn_shipped.ini" > "$D/run.ini"
cd "$D"
T0=$(date +%s.%N); "$R/crdmodel_amd/bin/FHNmodel_torus" run.ini | tail -4; echo "solver wall $(python -c "import time,sys; print(round(time.time()-float(sys.argv[1]),2))" $T0) s"
python - <<PY
import os, sys, time
sys.path.insert(0, "$R")
from crdmodel_amd import post
t0 = time.time()
post.main(["plot", "run.ini", "--dir", "."])
post.main(["map", "run.ini", "--dir", ".", "--mesh", "100"])
print("post-processing %.1f s" % (time.time() - t0))
PY
ls png | head -3; ls FHNstep | head -3
rm -f *.txt  # the text output (170 MB) stays on the box
