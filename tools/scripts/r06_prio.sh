R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd $R
bash tools/scripts/ab_env.sh "SATCV_M16P_PRIO=0" "SATCV_M16P_PRIO=10" "SATCV_M16P_PRIO=20" "SATCV_M16P_PRIO=30" "SATCV_M16P_PRIO=3" "SATCV_M16P_PRIO=2" "SATCV_M16P_PRIO=33" "SATCV_M16P_PRIO=22" "SATCV_M16P_PRIO=100" "SATCV_M16P_PRIO=133" > $O/r06_prio_step.txt 2>&1
cat $O/r06_prio_step.txt
