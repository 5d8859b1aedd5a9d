R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd $R
timeout 1500 python3 -m pytest tests/test_model_gpu.py tests/test_dp_gpu.py -x -q -m gpu -k "trajectory or fit or training_step or timed_configuration or data_parallel or dp or retrain or frozen or trainable" > $O/r06_eo_tests.log 2>&1
tail -4 $O/r06_eo_tests.log
bash tools/scripts/ab_env.sh "SATCV_EARLY_OPT=0" "SATCV_EARLY_OPT=1" > $O/r06_eo_step.txt 2>&1
cat $O/r06_eo_step.txt
