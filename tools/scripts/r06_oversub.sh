R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd $R
bash tools/scripts/ab_env.sh "SATCV_M16P_OVERSUB=1" "SATCV_M16P_OVERSUB=2" "SATCV_M16P_OVERSUB=4" > $O/r06_oversub_step.txt 2>&1
cat $O/r06_oversub_step.txt
