R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd $R
timeout 900 python3 -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py tests/test_lstm_gpu.py -x -q -m gpu -s -k "full_unet_training_step or pack or lstm_model_training_step" > $O/r06_run2_tests.log 2>&1
grep -n "cosine per tensor\|passed\|failed\|Error" $O/r06_run2_tests.log | cut -c1-3000 | head -20
bash tools/scripts/ab_env.sh "SATCV_WGRAD_LATE=1" "SATCV_WGRAD_LATE=0" > $O/r06_run2_step.txt 2>&1
cat $O/r06_run2_step.txt
python3 tools/timeline_quick.py 2>/dev/null | tail -5
