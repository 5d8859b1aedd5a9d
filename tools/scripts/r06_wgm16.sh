R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd $R
SATCV_WGRAD_M16=1 timeout 900 python3 -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "wgrad or full_unet_training_step or timed_configuration" > $O/r06_wgm16_tests.log 2>&1
tail -5 $O/r06_wgm16_tests.log
for rep in 1 2; do for v in 0 1; do echo "== SATCV_WGRAD_M16=$v"; SATCV_WGRAD_M16=$v python3 tools/wgrad_probe.py 2>&1 | grep " n64" | tail -10; done; done > $O/r06_wgm16_probe.txt 2>&1
cat $O/r06_wgm16_probe.txt
bash tools/scripts/ab_env.sh "SATCV_WGRAD_M16=0" "SATCV_WGRAD_M16=1" > $O/r06_wgm16_step.txt 2>&1
cat $O/r06_wgm16_step.txt
