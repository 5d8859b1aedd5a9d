# ablation builds of the persistent 16x16x32 kernel (build_m16p_variants.sh) on three shapes: usage r05_m16p_ablation.sh 1 2 4 ...
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd $R
SH="--opt igemm_m16=2 --shapes 64,128,128,64,128 64,64,64,128,128 64,32,32,256,256 --affine --reps 20"
echo "== product build"; python3 tools/conv_probe.py $SH | grep " n64"
for v in "$@"; do
  echo "== ablation bits $v (1 MFMA, 2 fragment reads, 4 weight DMA, 8 activation stream, 16 drain, 32 dump, 64 activation loads, 128 vmcnt wait)"
  SATCV_LIB=$R/satellite_computervision_amd/libsatcv_m16pABL$v.so python3 tools/conv_probe.py $SH | grep " n64"
done
echo "== product build again"; python3 tools/conv_probe.py $SH | grep " n64"
