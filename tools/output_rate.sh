#!/bin/bash
# Wall time of crd_run on an N x N FHN torus with 5 output intervals of ~400 steps each: text rows (the reference's format),
# text + .npy side-channel, .npy only -- how much of the output path hides behind the stepping.  tools/output_rate.sh [N]
N=${1:-4096}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d /tmp/crd_out.XXXXXX)
DT=$(python3 -c "import sys; sys.path.insert(0,'$ROOT'); import crdmodel_amd as c; p=c.make_params('fhn','torus',$N,80.0,20.0,0.12,1.25,ny=$N); print(0.8*c.stable_dt(p))")
TF=$(python3 -c "print(5*400*$DT)")
cat > $W/run.ini <<INI
[Parameters]
diffusion = 0.12
beta = 1.25
surfaceWidth = 20
surfaceLength = 80
waveLength = 0.1
waveWidth = 0.5
waveInside = 0
outputTimestep = 5
tBoundary = 0
tFinal = $TF
thetaMesh = $N
phiMesh = $N
betaMin = 0.7
betaMax = 1.7
[System]
includeAllVars = 0
varyBeta = 0
INI
for mode in "--binary-only" "--binary" ""; do
  mkdir -p $W/o; rm -f $W/o/*
  s=$(date +%s.%N)
  $ROOT/crdmodel_amd/bin/crd_run --model fhn --surface torus --quiet --outdir $W/o $mode $W/run.ini > /dev/null || echo "run failed"
  e=$(date +%s.%N)
  echo "N=$N mode='${mode:-text only}': $(python3 -c "print('%.2f s wall for 2000 steps + 6 frames' % ($e-$s))"), files: $(du -sh $W/o | cut -f1)"
done
rm -rf $W
