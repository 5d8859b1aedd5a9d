R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd $R
SH="--shapes 64,16,16,1024,512 64,32,32,512,256 64,64,64,128,128 --affine --reps 20"
echo "== product build"; python3 tools/conv_probe.py $SH
for v in "$@"; do
  echo "== ablation bits $v (1 fragment reads of steps 1-5, 2 staging, 4 interval wait + barrier, 8 MFMA, 16 epilogue)"
  SATCV_LIB=$R/satellite_computervision_amd/libsatcv_m16abl$v.so python3 tools/conv_probe.py $SH
done
echo "== product build again"; python3 tools/conv_probe.py $SH
