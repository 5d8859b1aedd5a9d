timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "persistent_16x16x32" 2>&1 | tail -5
R=$GRAFT_REPO_ROOT
SH="--opt igemm_m16=2 --shapes 64,128,128,64,128 64,128,128,128,128 64,64,64,128,128 64,64,64,128,256 64,64,64,256,128 64,32,32,256,256 --reps 20"
for aff in "" "--affine"; do
echo "== product $aff"; python3 tools/conv_probe.py $SH $aff | grep " n64"
echo "== m16p=0 $aff"; python3 tools/conv_probe.py --opt m16p=0 $SH $aff | grep " n64"
for v in PRIO2 ABL28 ABL16 ABL8 ABL1 ABL3; do echo "== $v $aff"; SATCV_LIB=$R/satellite_computervision_amd/libsatcv_m16p$v.so python3 tools/conv_probe.py $SH $aff | grep " n64"; done
done
