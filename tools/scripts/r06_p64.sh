R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd $R
bash tools/scripts/ab_env.sh "SATCV_M16P_PRIO64=30" "SATCV_M16P_PRIO64=333" "SATCV_M16P_PRIO64=33" "SATCV_M16P_PRIO64=222" "SATCV_M16P_PRIO64=0" > $O/r06_p64_step.txt 2>&1
cat $O/r06_p64_step.txt
