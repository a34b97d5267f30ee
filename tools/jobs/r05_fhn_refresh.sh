# Round 5, after the FHN kinetics lost a vector instruction per stage-point (EPSILON in vector registers): the headline kernel's records
# again -- the plan sweep with HBM traffic and launch durations, the issue-side counters.  (gpurun_out/r05/{sweep,sq}/)
set -x
R=$GRAFT_REPO_ROOT
bash $R/tools/jobs/r05_sweep.sh "fhn f64 8192" || exit 1
bash $R/tools/jobs/r05_sq.sh "fhn f64 8192" "0,1,1,1,1;1,0,1,1,2;0,1,1,1,2;0,2,1,1,2" || exit 1
