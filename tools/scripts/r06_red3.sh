R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd $R
timeout 1500 python3 -m pytest tests/test_model_gpu.py tests/test_dp_gpu.py -x -q -m gpu -k "trajectory or fit or training_step or timed_configuration or data_parallel or dp or deferred or siamese or atrous" > $O/r06_red3_tests.log 2>&1
tail -4 $O/r06_red3_tests.log
bash tools/scripts/ab_env.sh "SATCV_REDUCE_STREAM=0 SATCV_EARLY_OPT=0" "SATCV_REDUCE_STREAM=1 SATCV_EARLY_OPT=0" "SATCV_REDUCE_STREAM=1 SATCV_EARLY_OPT=1" "SATCV_REDUCE_STREAM=1 SATCV_EARLY_OPT=1 SATCV_WGRAD_WGS=112" "SATCV_REDUCE_STREAM=1 SATCV_EARLY_OPT=1 SATCV_WGRAD_WGS=96" "SATCV_REDUCE_STREAM=1 SATCV_EARLY_OPT=0 SATCV_WGRAD_WGS=112" > $O/r06_red3_step.txt 2>&1
cat $O/r06_red3_step.txt
