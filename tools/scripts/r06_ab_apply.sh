R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd $R
timeout 900 python3 -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -x -q -m gpu -s -k "full_unet_training_step or fused_backward or bn_bwd or 13_band_five or timed_configuration" > $O/r06_apply_tests.log 2>&1
grep -n "lowest cosine\|passed\|failed\|Error" $O/r06_apply_tests.log | head -20
for i in 1 2; do
for v in 0 1; do
SATCV_BN_APPLY=$v python3 tools/step_probe.py --only bn_bwd 2>&1 | grep TOTAL | sed "s/^/BN_APPLY=$v /"
done; done > $O/r06_ab_apply_probe.txt 2>&1
cat $O/r06_ab_apply_probe.txt
bash tools/scripts/ab_env.sh "SATCV_BN_APPLY=0" "SATCV_BN_APPLY=1" "SATCV_BN_APPLY=1 SATCV_FUSE_DGRAD_BN_BWD=2" "SATCV_BN_APPLY=1 SATCV_WGRAD_WGS=112" "SATCV_BN_APPLY=1 SATCV_WGRAD_WGS=144" "SATCV_BN_APPLY=1 SATCV_WGRAD_LATE=0" > $O/r06_ab_apply_step.txt 2>&1
cat $O/r06_ab_apply_step.txt
