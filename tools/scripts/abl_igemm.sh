# the deep 3x3 shapes per compile-time ablation build of the implicit-GEMM kernels (SATCV_ABLATE bits: see conv_igemm_fast.hip)
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd $R
for b in ${ABLS:-"" 8 72 4 16 2 1 32}; do
  if [ -z "$b" ]; then L=$R/satellite_computervision_amd/libsatcv.so; else L=$R/satellite_computervision_amd/libsatcvDSATCV_ABLATE$b.so; fi
  echo "== ABLATE ${b:-0}"
  SATCV_LIB=$L timeout 200 python3 tools/conv_probe.py --affine --shapes 64,16,16,1024,512 64,32,32,256,256 64,64,64,128,128 64,128,128,128,64 64,128,128,64,128 2>&1 | grep "k3"
done
