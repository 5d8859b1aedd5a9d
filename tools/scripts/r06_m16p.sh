R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd $R
timeout 900 python3 -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "m16 or persistent or conv2d or full_unet_training_step or fused_backward or timed_configuration or 13_band_five" > $O/r06_m16p_tests.log 2>&1
tail -5 $O/r06_m16p_tests.log
SH="64,128,128,64,128 64,64,64,128,128 64,64,64,128,256 64,64,64,256,128 64,32,32,256,256"
for rep in 1 2; do
  echo "== new"; python3 tools/conv_probe.py --opt igemm_m16=2 --shapes $SH 2>&1 | grep " n64"
  echo "== new --affine"; python3 tools/conv_probe.py --opt igemm_m16=2 --affine --shapes $SH 2>&1 | grep " n64"
done > $O/r06_m16p_probe.txt 2>&1
cat $O/r06_m16p_probe.txt
for s in "64,128,128,64,128" "64,64,64,128,256 --affine" "64,32,32,256,256"; do echo "== $s"; SATCV_LIB=$R/satellite_computervision_amd/libsatcv_m16pDSATCV_STAMP_M16P.so python3 tools/m16p_stamp_probe.py $s 2>&1 | grep -A12 "workgroup 0"; done > $O/r06_m16p_stamps.txt 2>&1
cat $O/r06_m16p_stamps.txt
timeout 300 python3 bench.py --steps 20 --warmup 5 --no-infer --no-cpu-baseline > $O/r06_m16p_bench.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('$O/r06_m16p_bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['extra']['kernel_ms_per_step'])"
