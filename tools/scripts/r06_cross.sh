R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd $R
bash tools/scripts/ab_env.sh "SATCV_EXP_CROSS=0" "SATCV_EXP_CROSS=1" "SATCV_EXP_CROSS=1 SATCV_WGRAD_WGS=112" "SATCV_EXP_CROSS=1 SATCV_WGRAD_WGS=96" "SATCV_EXP_CROSS=1 SATCV_WGRAD_WGS=64" > $O/r06_cross_step.txt 2>&1
cat $O/r06_cross_step.txt
