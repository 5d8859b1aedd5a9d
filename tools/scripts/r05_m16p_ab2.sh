# persistent 16x16x32 kernel: correctness, then product vs one-tile kernels vs ablation / priority variants (build_m16p_variants.sh) per shape
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "persistent_16x16x32" 2>&1 | tail -3
SH="--opt igemm_m16=2 --shapes 64,128,128,64,128 64,128,128,128,128 64,64,64,128,128 64,64,64,128,256 64,64,64,256,128 64,32,32,256,256 --reps 20"
for aff in "" "--affine"; do
echo "== product $aff"; python3 tools/conv_probe.py $SH $aff | grep " n64"
echo "== m16p=0 $aff"; python3 tools/conv_probe.py --opt m16p=0 $SH $aff | grep " n64"
for v in "$@"; do echo "== $v $aff"; SATCV_LIB=$R/satellite_computervision_amd/libsatcv_m16p$v.so python3 tools/conv_probe.py $SH $aff | grep " n64"; done
done
