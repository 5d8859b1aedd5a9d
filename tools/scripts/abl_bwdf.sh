# the fused backward on the three thin-layer shapes, per compile-time ablation build (see BWDF_ABL in conv_bwd_fused.hip)
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd $R
for b in "" 1 2 4 8 32 40; do
  if [ -z "$b" ]; then L=$R/satellite_computervision_amd/libsatcv.so; else L=$R/satellite_computervision_amd/libsatcvDBWDF_ABL$b.so; fi
  echo "== ABL ${b:-0}"
  SATCV_LIB=$L timeout 200 python3 tools/bwdf_probe.py --reps 20 2>&1 | grep -E "fused|not served" | head -8
done
