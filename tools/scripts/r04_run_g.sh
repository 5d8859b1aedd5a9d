O=gpurun_out
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -k "deeplab or invariance or acnn or residual" > $O/r04_t_g.txt 2>&1; tail -5 $O/r04_t_g.txt
for cfg in "SATCV_FUSE_RESIDUAL=1" "SATCV_FUSE_RESIDUAL=0" "SATCV_FUSE_RESIDUAL=1" "SATCV_FUSE_RESIDUAL=0"; do
  echo "== $cfg"; env $cfg timeout 200 python tools/deeplab_time.py 2>&1 | grep "^b"
done > $O/r04_dl_ab3.txt 2>&1
cat $O/r04_dl_ab3.txt
