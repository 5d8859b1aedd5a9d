R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd $R
timeout 900 python3 -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "persistent or full_unet_training_step or fused_backward or timed_configuration" > $O/r06_bn64_tests.log 2>&1
tail -5 $O/r06_bn64_tests.log
for rep in 1 2; do for v in 0 1; do
  echo "== BN64=$v"; SATCV_M16P_BN64=$v python3 tools/conv_probe.py --opt igemm_m16=2 --affine --shapes 64,128,128,128,64 64,64,64,128,64 64,64,64,256,64 2>&1 | grep " n64"
  echo "== BN64=$v (no affine)"; SATCV_M16P_BN64=$v python3 tools/conv_probe.py --opt igemm_m16=2 --shapes 64,128,128,128,64 64,64,64,128,64 2>&1 | grep " n64"
done; done > $O/r06_bn64_probe.txt 2>&1
cat $O/r06_bn64_probe.txt
bash tools/scripts/ab_env.sh "SATCV_M16P_BN64=0" "SATCV_M16P_BN64=1" > $O/r06_bn64_step.txt 2>&1
cat $O/r06_bn64_step.txt
