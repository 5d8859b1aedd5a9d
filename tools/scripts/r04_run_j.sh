O=gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -m gpu -q -k "deeplab or split_k or invariance or residual" > $O/r04_t_j.txt 2>&1; tail -4 $O/r04_t_j.txt
for cfg in "SATCV_DEEPLAB_SPLITK=1" "SATCV_DEEPLAB_SPLITK=0" "SATCV_DEEPLAB_SPLITK=1" "SATCV_DEEPLAB_SPLITK=0"; do
  echo "== $cfg"; env $cfg timeout 200 python tools/deeplab_time.py 2>&1 | grep "^b"
done > $O/r04_dl_ab5.txt 2>&1
cat $O/r04_dl_ab5.txt
