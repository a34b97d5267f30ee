# Round 5, after the Goldbeter fp64 two-step kernel became the block-as-the-strip form: its records again -- the plan sweep with HBM
# traffic and the issue-side counters.  (gpurun_out/r05/{sweep,sq}/)
set -x
R=$GRAFT_REPO_ROOT
bash $R/tools/jobs/r05_sweep.sh "goldbeter f64 4096" || exit 1
bash $R/tools/jobs/r05_sq.sh "goldbeter f64 4096" "0,0,1,1,1;1,0,1,1,2;1,1,1,1,2;0,2,1,1,2" || exit 1
