R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd $R
bash tools/scripts/ab_env.sh "SATCV_WGRAD_WGS=128" "SATCV_WGRAD_WGS=120" "SATCV_WGRAD_WGS=136" "SATCV_WGRAD_WGS=128 SATCV_EW_PER_CU=4" "SATCV_WGRAD_WGS=128 SATCV_EW_PER_CU=8" > $O/r06_wgs_step.txt 2>&1
cat $O/r06_wgs_step.txt
