O=gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q -k "aspp or deeplab or siamese or dilat or acnn or atrous or strid" > $O/r04_t_i.txt 2>&1; tail -5 $O/r04_t_i.txt
for cfg in "SATCV_DB_TL=1" "SATCV_DB_TL=0" "SATCV_DB_TL=1" "SATCV_DB_TL=0" "SATCV_DB_TL=192"; do
  echo "== $cfg"; env $cfg timeout 200 python tools/deeplab_time.py 2>&1 | grep "^b"
done > $O/r04_dl_ab4.txt 2>&1
cat $O/r04_dl_ab4.txt
