O=gpurun_out
for cfg in "SATCV_INFER_GRAPH=1 SATCV_SPLITK=0" "SATCV_INFER_GRAPH=0 SATCV_SPLITK=0" "SATCV_INFER_GRAPH=1 SATCV_SPLITK=1" "SATCV_INFER_GRAPH=0 SATCV_SPLITK=1" "SATCV_INFER_GRAPH=1 SATCV_SPLITK=0 SATCV_SPLITK_TL=0"; do
  echo "== $cfg"; env $cfg timeout 200 python tools/deeplab_time.py 2>&1 | grep "^b"
done > $O/r04_dl_ab.txt 2>&1
cat $O/r04_dl_ab.txt
bash tools/scripts/deeplab_trace.sh > $O/r04_dl_trace.txt 2>&1; head -24 $O/r04_dl_trace.txt | cut -c1-200
