R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd $R
timeout 600 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "m16 or persistent or conv2d" > $O/r06_m16p_tests.log 2>&1
tail -3 $O/r06_m16p_tests.log
SH="64,128,128,64,128 64,64,64,128,128 64,64,64,128,256 64,64,64,256,128 64,32,32,256,256 64,32,32,256,512"
for rep in 1 2; do
  for L in "" "$R/satellite_computervision_amd/libsatcv_m16pold.so"; do
    echo "== lib=${L:-new}"; SATCV_LIB=$L python3 tools/conv_probe.py --opt igemm_m16=2 --shapes $SH 2>&1 | grep " n64"
    echo "== lib=${L:-new} --affine"; SATCV_LIB=$L python3 tools/conv_probe.py --opt igemm_m16=2 --affine --shapes $SH 2>&1 | grep " n64"
  done
done > $O/r06_m16p_ab_probe.txt 2>&1
cat $O/r06_m16p_ab_probe.txt
for s in "64,128,128,64,128" "64,64,64,128,256 --affine" "64,32,32,256,256"; do echo "== $s"; SATCV_LIB=$R/satellite_computervision_amd/libsatcv_m16pDSATCV_STAMP_M16P.so python3 tools/m16p_stamp_probe.py $s 2>&1 | grep -A12 "workgroup 0"; done > $O/r06_m16p_stamps.txt 2>&1
grep "wave  [0489]" $O/r06_m16p_stamps.txt
for rep in 1 2; do for L in "" "$R/satellite_computervision_amd/libsatcv_m16pold.so"; do
  out=$(SATCV_LIB=$L python3 bench.py --steps 20 --warmup 5 --no-infer --no-cpu-baseline 2>/dev/null | tail -1)
  echo "lib=${L:-new} :: $(echo "$out" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'tiles/s frac', d['roofline']['frac'], d['extra']['kernel_ms_per_step'])")"
done; done > $O/r06_m16p_ab_step.txt 2>&1
cat $O/r06_m16p_ab_step.txt
