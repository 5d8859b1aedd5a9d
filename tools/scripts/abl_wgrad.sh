# the weight-gradient launches of the step per compile-time ablation build (SATCV_WABLATE bits: see conv_wgrad.hip)
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd $R
for b in "" 1 2 4 8 9; do
  if [ -z "$b" ]; then L=$R/satellite_computervision_amd/libsatcv.so; else L=$R/satellite_computervision_amd/libsatcvDSATCV_WABLATE$b.so; fi
  echo "== WABLATE ${b:-0}"
  SATCV_LIB=$L timeout 200 python3 tools/step_probe.py --only "wgrad k3" 2>&1 | grep -E "wgrad"
done
