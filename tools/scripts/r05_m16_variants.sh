R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd $R
SH="--shapes 64,16,16,1024,512 64,32,32,512,256 64,64,64,128,128 --reps 20"
for rep in 1 2; do
echo "== product build, affine"; python3 tools/conv_probe.py $SH --affine
for v in "$@"; do
  echo "== variant $v, affine"; SATCV_LIB=$R/satellite_computervision_amd/libsatcv_m16$v.so python3 tools/conv_probe.py $SH --affine
done
done
echo "== product build, no affine"; python3 tools/conv_probe.py $SH
for v in "$@"; do
  echo "== variant $v, no affine"; SATCV_LIB=$R/satellite_computervision_amd/libsatcv_m16$v.so python3 tools/conv_probe.py $SH
done
