O=gpurun_out
for cfg in "SATCV_DB1X1_SMALL=1" "SATCV_DB1X1_SMALL=0" "SATCV_DB1X1_SMALL=1" "SATCV_DB1X1_SMALL=0"; do
  echo "== $cfg"; env $cfg timeout 200 python tools/deeplab_time.py 2>&1 | grep "^b"
done > $O/r04_dl_ab2.txt 2>&1
cat $O/r04_dl_ab2.txt
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -k "deeplab or invariance" > $O/r04_t_f.txt 2>&1; tail -3 $O/r04_t_f.txt
